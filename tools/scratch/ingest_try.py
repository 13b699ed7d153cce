import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from mod16_amd import _lib
from mod16_amd.raster import RasterEngine
from mod16_amd.utils import restore_bplut, bplut_table
from mod16_amd.models import COLLECTION61_BPLUT
from oracle import mod16_oracle as oracle
table = bplut_table(restore_bplut(COLLECTION61_BPLUT), beta=250)
bplut = {k: table[:, j] for j, k in enumerate(oracle.PARAM_NAMES)}
n = (21600 * 43200 // 32) // 8192 * 8192
for math in (_lib.MATH_FAST, _lib.MATH_MIXED):
    print(json.dumps(bench.ingest_raw_series(torch, np, _lib, RasterEngine, table, 0, math, n, 46, bplut)), flush=True)

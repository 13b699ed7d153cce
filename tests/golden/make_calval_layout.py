#!/usr/bin/env python3
"""Build-container script: the Cal-Val HDF5 layout as the REFERENCE states it, written as data to
tests/golden/calval_layout.json (names, ranks, which names its configuration may change) --

  * `documented`: the file specification in the module docstring of mod16/calibration.py
    (lines 50-112): every dataset path, its dimensions (T time steps, N towers, P sub-grid
    pixels, L land-cover types, Y years) and whether it or its group carries the star that marks
    a configurable name;
  * `load_data`: what Calibration._load_data (lines 304-423) actually opens: literal dataset
    paths, `lookup[KEY]` / `lookup[KEY][i]` reads (KEY -> the indices used; [] = the whole entry)
    and configuration keys that name a dataset;
  * `shipped_config_datasets`: the `data: datasets:` mapping of the configuration file the
    reference ships (mod16/data/MOD16_calibration_config.yaml) -- the keys a user is expected to set.

tests/test_h5_to_store.py checks tools/h5_to_store.py's field map against this file. Only text of
the reference is read (nothing is imported); the output holds names and shapes, no source text.

    python tests/golden/make_calval_layout.py [/root/reference]
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def documented(lines):
    out, group, group_star, pending = [], '', False, None
    for line in lines:
        m = re.match(r'^    (\*?)([A-Za-z0-9_\-]+)/\s*$', line)
        if m:
            group, group_star = m.group(2), bool(m.group(1))
            continue
        m = re.match(r'^    (\*?)([A-Za-z0-9_]+)\s+-- \(([^)]*)\)', line)      # top-level dataset (time, weights)
        if m and not line.startswith('      '):
            out.append({'path': m.group(2), 'dims': [d.strip() for d in m.group(3).split('x')],
                        'starred': bool(m.group(1)), 'group_starred': False})
            continue
        m = re.match(r'^      (\*?)([A-Za-z0-9_]+)\s*(?:-- \(([^)]*)\))?', line)
        if m and line.startswith('      ') and not line.startswith('       '):
            entry = {'path': '%s/%s' % (group, m.group(2)), 'starred': bool(m.group(1)), 'group_starred': group_star}
            if m.group(3) is not None:
                entry['dims'] = [d.strip() for d in m.group(3).split('x')]
                out.append(entry)
            else:
                pending = entry              # the dimensions follow on the next line
            continue
        m = re.match(r'^\s+-- \(([^)]*)\)', line)
        if m and pending is not None:
            pending['dims'] = [d.strip() for d in m.group(1).split('x')]
            out.append(pending)
            pending = None
    return out


def load_data(text):
    literal = sorted(set(re.findall(r"hdf\['([^']+)'\]", text)))
    lookups = {}
    for key, idx in re.findall(r"lookup\['(\w+)'\](?:\[(\d)\])?", text):
        lookups.setdefault(key, set())
        if idx:
            lookups[key].add(int(idx))
    config = sorted(set(re.findall(r"hdf\[self\.config\['data'\]\['(\w+)'\]\]", text)))
    return {'literal_paths': literal, 'lookup_keys': {k: sorted(v) for k, v in sorted(lookups.items())},
            'config_keys': config}


def shipped_config(path):
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f)
    return cfg['data']['datasets'], {k: cfg['data'][k] for k in ('class_map', 'target_observable')}


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    src = open(os.path.join(ref, 'mod16', 'calibration.py')).read().split('\n')
    datasets, other = shipped_config(os.path.join(ref, 'mod16', 'data', 'MOD16_calibration_config.yaml'))
    out = {
        'source': 'arthur-e/MOD16 mod16/calibration.py: docstring lines 50-112, _load_data lines 304-423; '
                  'mod16/data/MOD16_calibration_config.yaml',
        'documented': documented(src[49:112]),
        'load_data': load_data('\n'.join(src[303:423])),
        'shipped_config_datasets': datasets, 'shipped_config_other': other,
    }
    with open(os.path.join(HERE, 'calval_layout.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write('\n')
    print('%d documented datasets, %d lookup keys' % (len(out['documented']), len(out['load_data']['lookup_keys'])))


if __name__ == '__main__':
    main()

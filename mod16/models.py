'''``mod16.models`` of the MI355X build (see ``mod16/__init__.py``).'''
from mod16_amd.models import *   # noqa: F401,F403
from mod16_amd.models import MOD16Collection61, PFT_ALL  # noqa: F401

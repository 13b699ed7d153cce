'''``mod16.utils`` of the MI355X build (see ``mod16/__init__.py``).'''
from mod16_amd.utils import *    # noqa: F401,F403
from mod16_amd.utils import BPLUT_FIELD_LOOKUP, restore_bplut, write_bplut, pft_dominant  # noqa: F401

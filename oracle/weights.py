"""TEST INFRASTRUCTURE: the seeded checkpoint generator lives in the package (data only, shared with bench.py);
re-exported here for the oracle-side scripts and tests."""
from advancedliteratemachinery_amd.utils.synthetic import *  # noqa: F401,F403
from advancedliteratemachinery_amd.utils.synthetic import SWIN_B, make_state_dict, relative_position_index, state_dict_spec  # noqa: F401

"""Build an A/B or probe variant of the library: libevflow_<name>.so with extra -D flags for some sources, every other object
taken from the main build; loaded through EVF_LIB=<path> (measurements only, never the product path).

    python tools/ab_variant.py ft_nomfma evf_fwd_teams.hip:-DFT_PROBE_NOMFMA evf_fwd_teams.hip:-DFT_PF=3
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_flow_amd import build  # noqa: E402

name, defs = sys.argv[1], collections.defaultdict(list)
for a in sys.argv[2:]:
    f, d = a.split(":", 1)
    defs[f].append(d)
print(build.build_variant(name, dict(defs)))

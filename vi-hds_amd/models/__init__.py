"""Model registry: the same string keys as the reference's models/__init__.py:19-35, so `model:` in a spec
YAML selects the same model; every key maps to a kernel model.  debug_constant, inducer_*, relay_* and degrader_* cannot run
in the reference itself (its classes raise at construction / call).  relay / degrader / inducer / prpr
`_precisions` -- the only specs of those models the reference ships -- are pinned against the MODIFIED reference (the two
construction defects repaired in memory, tests/golden/make_fixtures.py --patched; tests/test_config5_parity.py,
tests/test_general_tail.py); their constant-precision forms share that RHS code; debug_constant is unpinned."""
from models import (auto_constant, debug, degrader_constant, dr_blackbox, dr_constant, inducer_constant, prpr_constant,
                    relay_constant)


LOOKUP = {
    "debug_constant": debug.Debug_Constant,
    "auto_constant": auto_constant.Auto_Constant,
    "auto_constant_precisions": auto_constant.Auto_Constant_Precisions,
    "degrader_constant": degrader_constant.Degrader_Constant,
    "degrader_constant_precisions": degrader_constant.Degrader_Constant_Precisions,
    "dr_constant": dr_constant.DR_Constant,
    "dr_constant_v2": dr_constant.DR_Constant_V2,
    "dr_constant_precisions": dr_constant.DR_Constant_Precisions,
    "dr_constant_precisions_v2": dr_constant.DR_Constant_Precisions_V2,
    "dr_blackbox": dr_blackbox.DR_Blackbox,
    "inducer_constant": inducer_constant.Inducer_Constant,
    "inducer_constant_precisions": inducer_constant.Inducer_Constant_Precisions,
    "prpr_constant": prpr_constant.PRPR_Constant,
    "prpr_constant_precisions": prpr_constant.PRPR_Constant_Precisions,
    "relay_constant": relay_constant.Relay_Constant,
    "relay_constant_precisions": relay_constant.Relay_Constant_Precisions,
}

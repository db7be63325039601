"""mc_pilco_amd -- MI355X-native hot path of MC-PILCO (particle rollout + GP dynamics).

The directory is named ``mc-pilco_amd``; ``mcp_boot`` registers it as ``mc_pilco_amd``.
Sub-packages ``gpr_lib``, ``model_learning`` and ``policy_learning`` mirror the reference's
module paths and class names; ``hipabi`` is the ctypes binding of the C-ABI library built
from ``csrc/`` (``include/mcpilco_hip.h``).
"""
__version__ = "0.1.0"

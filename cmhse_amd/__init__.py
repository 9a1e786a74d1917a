"""cmhse_amd: MI355X-native (gfx950) implementation of CMHSE's embedding-and-ranking hot path.

Host-side mirror of the reference's Python interface (model / layers / loss / evaluation,
/root/reference/{model,layers,loss,evaluation}.py) over a C-ABI HIP library
(`cmhse_amd/csrc`, declared in `include/cmhse_hip.h`).  Importing the package does not load the
HIP library; the first operator call does, and fails loudly if it is missing.
"""

__version__ = '0.1.0'

"""Drop-in shim: lets a train.py-style driver written against the reference keep its
`from model import ...` lines (train.py:9-10).  Put this directory first on sys.path
(see INTEGRATION.md); everything is re-exported from cmhse_amd.model."""
from cmhse_amd.model import *  # noqa: F401,F403

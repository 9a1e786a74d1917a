"""Drop-in shim: lets a train.py-style driver written against the reference keep its
`from evaluation import ...` lines (train.py:9-10).  Put this directory first on sys.path
(see INTEGRATION.md); everything is re-exported from cmhse_amd.evaluation."""
from cmhse_amd.evaluation import *  # noqa: F401,F403

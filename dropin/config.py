"""Drop-in shim: put this folder on sys.path (or cd into it) and the reference's `from config import ...` resolves
to the MI355X implementation (svs_unet_pytorch_amd.config)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svs_unet_pytorch_amd.config import *  # noqa: F401,F403,E402

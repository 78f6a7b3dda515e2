"""Drop-in shim: `python data.py <the reference's flags>` runs the MI355X implementation (svs_unet_pytorch_amd.data)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svs_unet_pytorch_amd.data import main  # noqa: E402

if __name__ == "__main__":
    main()

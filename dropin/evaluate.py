"""`python evaluate.py ...` drop-in: the reference's evaluate.py command line on this package's BSS-eval."""
from svs_unet_pytorch_amd.evaluate import *  # noqa: F401,F403
from svs_unet_pytorch_amd.evaluate import main

if __name__ == "__main__":
    main()

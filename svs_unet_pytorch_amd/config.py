"""Constants of the pipeline (same names and values as the reference's config.py:47-51, which its
data.py / train.py / inference.py star-import) and the zero-padded index formatter (config.py:1-9)."""

WINDOW_SIZE = 1024      # STFT window = n_fft            (config.py:47)
HOP_SIZE = 768          # STFT hop                        (config.py:48)
SAMPLE_RATE = 8192      # resampling rate of data.py      (config.py:49)
INPUT_LEN = 128         # frames per network tile         (config.py:50)
SAMPLES_PER_SONG = 64   # random crops per song per epoch (config.py:51)


def num2str(n):
    """4-digit zero-padded decimal, longer numbers unchanged (config.py:1-9)."""
    return str(n).rjust(4, "0")

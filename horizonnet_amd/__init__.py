"""horizonnet_amd -- MI355X (gfx950) engine for the HorizonNet hot path.

Drop-in surface (same names / arguments / error behaviour as the reference):

* ``HorizonNet(backbone, use_rnn)``            -- reference ``model.py:185-281``
* ``pano_stretch(img, corners, kx, ky)``        -- reference ``misc/panostretch.py:81-117``
* ``find_N_peaks(signal, r, min_v, N)``         -- reference ``inference.py:21-29``
* ``inference(net, x, device, ...)``            -- reference ``inference.py:65-141`` (+ ``inference_batch``)
* ``postproc``                                  -- reference ``misc/post_proc.py`` (host numpy, as in the reference)

All compute runs in hand-written HIP kernels behind the C ABI declared in
``include/horizonnet_hip.h`` (``libhorizonnet_hip.so``, built in-tree by
``horizonnet_amd/csrc/build.sh``).  There is no CPU or eager-PyTorch fallback:
importing the compute entry points without the library, or calling them with
host tensors, raises.
"""
from .model import HorizonNet  # noqa: F401
from .panostretch import pano_stretch, pano_stretch_batch  # noqa: F401
from .peaks import find_N_peaks, find_peaks_batch  # noqa: F401
from . import postproc  # noqa: F401
from .inference import inference, inference_batch, inference_stream  # noqa: F401
from .parallel import allreduce_mean_, broadcast_module_  # noqa: F401
from .optim import FusedAdam  # noqa: F401

__all__ = ["HorizonNet", "pano_stretch", "pano_stretch_batch", "find_N_peaks", "find_peaks_batch", "inference", "inference_batch", "inference_stream", "postproc",
           "allreduce_mean_", "broadcast_module_", "FusedAdam"]

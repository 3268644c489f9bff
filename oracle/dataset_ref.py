"""CPU restatement of the image half of reference ``dataset.PanoCorBonDataset.__getitem__`` (``dataset.py:52,82,
88-89,95-96,100-104,123``) -- TEST INFRASTRUCTURE ONLY (the checker for ``hn_augment_batch``).

``gen_golden.py dataset`` checks it bit for bit against the unmodified reference on the committed synthetic dataset."""
import numpy as np

from . import panostretch_ref


def augment_image(img_u8, kx=1.0, ky=1.0, flip=0, roll=None, gamma=1.0):
    """img_u8 [H,W,3] uint8 -> float32 [3,H,W]; `roll` None = rotation augmentation off."""
    img = np.array(img_u8, np.float32)[..., :3] / 255.
    if not (kx == 1.0 and ky == 1.0):
        img = panostretch_ref.pano_stretch(img, np.zeros((1, 2)), kx, ky)[0]    # misc/panostretch.py:91-102
    if flip:
        img = np.flip(img, axis=1)
    if roll is not None:
        img = np.roll(img, roll, axis=1)
    if gamma != 1.0:
        img = img ** gamma                                              # float32 ** python float -> float32 powf
    return np.ascontiguousarray(img.transpose([2, 0, 1]))

"""The reference's per-item augmentation stack (dataloading/dataset.py:171-205) restated in numpy / scipy.

The reference builds it from albumentations classes with their DEFAULT parameters; albumentations / volumentations are not
installed in this image and cannot be fetched, so nothing here can be compared with them: **parity unpinned**.  What IS taken
from the reference's source text: the structure (three `OneOf` groups applied with p = 0.3 / 0.35 / 0.4 inside an always-on
`Compose`, then a `Compose(p=0.5)` holding one `CoarseDropout3D`), the explicit CoarseDropout3D arguments, and the calling
convention -- the 3-D patch (Z, Y, X) is handed over as `image=`, i.e. albumentations sees an image of height Z, width Y with X
CHANNELS: every 2-D operation below therefore acts in the (Z, Y) plane with one parameter draw for all x.  The members'
parameter ranges are albumentations' documented defaults as far as they are public knowledge (each function says which);
members of a `OneOf` all carry the default p = 0.5, so the choice among them is uniform, and a transform without an explicit p
fires with its default 0.5.

Host-side numpy work in the DataLoader workers, like the reference's; nothing here touches the GPU path."""
import numpy as np
import torch
from scipy import ndimage

_rng = None


def _generator():
    """one generator per process, seeded from torch's per-worker seed (DataLoader gives every worker its own)"""
    global _rng
    if _rng is None:
        _rng = np.random.default_rng(torch.initial_seed() % (1 << 63))
    return _rng


def _clip(a):
    return np.clip(a, 0.0, 1.0, out=a)


# ---- group 1: brightness -------------------------------------------------------------------------------------------------
def random_brightness_contrast(img, rng):
    """RandomBrightnessContrast: brightness_limit = contrast_limit = (-0.2, 0.2), brightness_by_max (max = 1 for float images)"""
    alpha = 1.0 + rng.uniform(-0.2, 0.2)
    beta = rng.uniform(-0.2, 0.2)
    return _clip(img * np.float32(alpha) + np.float32(beta))


def illumination(img, rng):
    """Illumination: mode "linear", intensity_range (0.01, 0.2), effect_type "both", angle_range (0, 360): a linear ramp across
    the (Z, Y) plane that brightens or darkens by up to `intensity`"""
    h, w = img.shape[:2]
    intensity = rng.uniform(0.01, 0.2) * (1.0 if rng.random() < 0.5 else -1.0)
    ang = np.deg2rad(rng.uniform(0.0, 360.0))
    yy, xx = np.meshgrid(np.linspace(0.0, 1.0, h), np.linspace(0.0, 1.0, w), indexing="ij")
    g = xx * np.cos(ang) + yy * np.sin(ang)
    g = (g - g.min()) / max(g.max() - g.min(), 1e-12)
    factor = (1.0 + intensity * g).astype(np.float32)
    return _clip(img * factor[:, :, None])


# ---- group 2: noise ------------------------------------------------------------------------------------------------------
def multiplicative_noise(img, rng):
    """MultiplicativeNoise: multiplier (0.9, 1.1), per_channel False, elementwise False -> one factor for the patch"""
    return _clip(img * np.float32(rng.uniform(0.9, 1.1)))


def gauss_noise(img, rng):
    """GaussNoise: std_range (0.2, 0.44) of the value range, mean 0, per_channel True -> independent noise per element"""
    sigma = rng.uniform(0.2, 0.44)
    return _clip(img + rng.normal(0.0, sigma, size=img.shape).astype(np.float32))


# ---- group 3: blur / resolution ------------------------------------------------------------------------------------------
def _filter_plane(img, kernel):
    """2-D correlation in the (Z, Y) plane, the same kernel for every x; border reflect-101 (cv2's default) = scipy "mirror\""""
    return ndimage.correlate(img, kernel[:, :, None].astype(np.float32), mode="mirror")


def motion_blur(img, rng):
    """MotionBlur: blur_limit (3, 7): a normalised line through a k x k kernel at a random angle (the centre shifts and the
    direction bias of the newer releases are not modelled)"""
    k = int(rng.choice([3, 5, 7]))
    ang = np.deg2rad(rng.uniform(0.0, 360.0))
    kern = np.zeros((k, k), np.float32)
    c = (k - 1) / 2.0
    for t in np.linspace(-c, c, 4 * k):
        kern[int(round(c + t * np.sin(ang))), int(round(c + t * np.cos(ang)))] = 1.0
    return _clip(_filter_plane(img, kern / kern.sum()))


def defocus(img, rng):
    """Defocus: radius (3, 10), alias_blur (0.1, 0.5): a disc of that radius, its edge softened by a Gaussian of sigma alias_blur"""
    r = int(rng.integers(3, 11))
    alias = rng.uniform(0.1, 0.5)
    ax = np.arange(-r, r + 1)
    yy, xx = np.meshgrid(ax, ax, indexing="ij")
    disc = ((yy * yy + xx * xx) <= r * r).astype(np.float32)
    disc = ndimage.gaussian_filter(disc, alias, mode="constant")
    return _clip(_filter_plane(img, disc / disc.sum()))


def downscale(img, rng):
    """Downscale: scale_range (0.25, 0.25), nearest-neighbour both ways (cv2.INTER_NEAREST: source index = floor(dst / scale))"""
    h, w = img.shape[:2]
    hs, ws = max(1, int(round(h * 0.25))), max(1, int(round(w * 0.25)))
    down_r = np.minimum((np.arange(hs) * (h / hs)).astype(np.int64), h - 1)
    down_c = np.minimum((np.arange(ws) * (w / ws)).astype(np.int64), w - 1)
    small = img[down_r][:, down_c]
    up_r = np.minimum((np.arange(h) * (hs / h)).astype(np.int64), hs - 1)
    up_c = np.minimum((np.arange(w) * (ws / w)).astype(np.int64), ws - 1)
    return np.ascontiguousarray(small[up_r][:, up_c])


def advanced_blur(img, rng):
    """AdvancedBlur: blur_limit (3, 7), sigma_x / sigma_y (0.2, 1.0), rotate (-90, 90), beta (0.5, 8), noise (0.9, 1.1): a
    rotated generalised-Gaussian kernel exp(-(q^beta) / 2), q the anisotropic squared radius, with multiplicative kernel noise"""
    k = int(rng.choice([3, 5, 7]))
    sx, sy = rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0)
    ang = np.deg2rad(rng.uniform(-90.0, 90.0))
    beta = rng.uniform(0.5, 8.0)
    ax = np.arange(k) - (k - 1) / 2.0
    yy, xx = np.meshgrid(ax, ax, indexing="ij")
    xr = xx * np.cos(ang) + yy * np.sin(ang)
    yr = -xx * np.sin(ang) + yy * np.cos(ang)
    q = (xr / sx) ** 2 + (yr / sy) ** 2
    kern = np.exp(-0.5 * np.power(q, beta)) * rng.uniform(0.9, 1.1, size=(k, k))
    return _clip(_filter_plane(img, (kern / kern.sum()).astype(np.float32)))


# ---- volumetric ----------------------------------------------------------------------------------------------------------
def coarse_dropout_3d(vol, rng, fill=0.5, num_holes_range=(1, 4), depth_range=(0.1, 0.4), height_range=(0.1, 0.4),
                      width_range=(0.1, 0.4)):
    """CoarseDropout3D with the arguments the reference passes (dataset.py:191-197): 1-4 boxes, each 10-40 % of the patch along
    every axis, placed uniformly inside it, filled with 0.5"""
    out = vol.copy()
    D, H, W = vol.shape[:3]
    for _ in range(int(rng.integers(num_holes_range[0], num_holes_range[1] + 1))):
        d = max(1, int(D * rng.uniform(*depth_range)))
        h = max(1, int(H * rng.uniform(*height_range)))
        w = max(1, int(W * rng.uniform(*width_range)))
        z0, y0, x0 = (int(rng.integers(0, D - d + 1)), int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1)))
        out[z0:z0 + d, y0:y0 + h, x0:x0 + w] = fill
    return out


GROUPS = (
    (0.30, (random_brightness_contrast, illumination)),
    (0.35, (multiplicative_noise, gauss_noise)),
    (0.40, (motion_blur, defocus, downscale, advanced_blur)),
)
P_VOLUME_COMPOSE, P_COARSE_DROPOUT = 0.5, 0.5


def augment_image(img, rng=None):
    """the whole stack on one float32 patch in [0, 1]: (Z, Y, X), or (C, Z, Y, X) with every channel treated alike (one draw).
    Targets are never touched (the reference passes only `image` / `volume`, dataset.py:200-205).  Returns a new array."""
    rng = rng if rng is not None else _generator()
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 4:
        state = rng.bit_generator.state
        outs = []
        for c in range(img.shape[0]):
            rng.bit_generator.state = state            # the same draws for every channel of one patch
            outs.append(augment_image(img[c], rng))
        return np.stack(outs)
    out = img.copy()
    for p, members in GROUPS:
        if rng.random() < p:
            out = members[int(rng.integers(len(members)))](out, rng)
    if rng.random() < P_VOLUME_COMPOSE and rng.random() < P_COARSE_DROPOUT:
        out = coarse_dropout_3d(out, rng)
    return np.ascontiguousarray(out, dtype=np.float32)

"""Film helpers on the host side of the C ABI: XYZ/weight film -> RGB image exactly as
ImageFilm::WriteImage does it (film/image.cpp:178-213), PFM I/O (core/imageio.cpp:419-446 —
the format the reference writes without OpenEXR) and the image-difference metric of
tools/exrdiff.cpp:76-108 (MSE = sum(d^2)/(3*W*H), RMSE = sqrt(MSE))."""
import numpy as np


def xyzw_to_rgb(film):
    """film: (H, W, 4) float32 {X, Y, Z, weightSum} -> (H, W, 3) float32 RGB.
    XYZToRGB (core/spectrum.h:51-55), divide by weightSum where non-zero, clamp at 0.
    float32 arithmetic in the reference's order."""
    f = np.asarray(film, dtype=np.float32)
    x, y, z, w = f[..., 0], f[..., 1], f[..., 2], f[..., 3]
    c = np.float32
    r = c(3.240479) * x - c(1.537150) * y - c(0.498535) * z
    g = c(-0.969256) * x + c(1.875991) * y + c(0.041556) * z
    b = c(0.055648) * x - c(0.204043) * y + c(1.057311) * z
    rgb = np.stack([r, g, b], axis=-1).astype(np.float32)
    nz = w != 0
    inv = np.zeros_like(w)
    inv[nz] = c(1.0) / w[nz]
    scaled = np.maximum(rgb * inv[..., None], c(0.0))
    rgb = np.where(nz[..., None], scaled, rgb)
    return rgb.astype(np.float32)


def read_pfm(path):
    with open(path, "rb") as f:
        kind = f.readline().strip()
        if kind not in (b"PF", b"Pf"):
            raise ValueError("not a PFM file")
        w, h = map(int, f.readline().split())
        scale = float(f.readline().strip())
        nc = 3 if kind == b"PF" else 1
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4", count=w * h * nc)
    img = data.reshape(h, w, nc).astype(np.float32)
    return img[::-1].copy()  # PFM stores bottom-to-top


def write_pfm(path, img):
    img = np.asarray(img, dtype="<f4")
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(img[::-1].tobytes())


def rmse(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))

import importlib

import numpy as np

film = importlib.import_module("pbrt-v2_amd.film")


def test_xyz_to_rgb_and_weights():
    f = np.zeros((2, 2, 4), np.float32)
    f[0, 0] = [0.412453 + 0.357580 + 0.180423, 0.212671 + 0.715160 + 0.072169, 0.019334 + 0.119193 + 0.950227, 1]
    f[0, 1] = 2 * f[0, 0]
    f[1, 0] = [-1, -1, -1, 1]
    img = film.xyzw_to_rgb(f)
    assert np.allclose(img[0, 0], 1, atol=1e-5) and np.allclose(img[0, 1], 1, atol=1e-5)
    assert (img[1, 0] == 0).all() and (img[1, 1] == 0).all()


def test_pfm_roundtrip_and_rmse(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.random((5, 7, 3)).astype(np.float32)
    p = tmp_path / "a.pfm"
    film.write_pfm(p, a)
    b = film.read_pfm(p)
    assert np.array_equal(a, b)
    assert film.rmse(a, b) == 0
    assert abs(film.rmse(a, a + 0.5) - 0.5) < 1e-6

"""CPU tests of the C++ host pieces (srmeetsps-cuda_amd/host/): MAT5 reader/writer against scipy.io,
PNG decoder against PIL, the image-folder loader, and the depth pre-processing against independent
numpy restatements of the published definitions (OpenCV itself is not available: parity unpinned)."""
import os
import subprocess

import numpy as np
import pytest
import scipy.io

f32 = np.float32


@pytest.fixture(scope="module")
def host(pkg):
    pkg.host.build()
    pkg.host.load()
    return pkg.host


def _cm(a2d):
    return np.ascontiguousarray(np.asarray(a2d).T).reshape(-1)


@pytest.mark.parametrize("compress", [False, True])
def test_mat5_reader_matches_scipy(host, tmp_path, compress):
    rng = np.random.default_rng(0)
    h, w, c, n, sf = 8, 12, 3, 4, 2
    I = rng.uniform(size=(h, w, c, n)); K = np.array([[100.0, 0, 5.5], [0, 101.0, 3.5], [0, 0, 1]])
    mask = (rng.uniform(size=(h, w)) > 0.3); z0 = rng.uniform(1, 2, size=(h // sf, w // sf, 2))
    path = str(tmp_path / "d.mat")
    scipy.io.savemat(path, {"I": I, "K": K, "mask": mask, "sf": float(sf), "z0": z0}, do_compression=compress)
    dh = host.load_dataset("matlab", path, preprocess=False)
    assert (dh.I_h, dh.I_w, dh.I_c, dh.I_n, dh.sf, dh.z0_n) == (h, w, c, n, sf, 2)
    np.testing.assert_array_equal(dh.I, np.transpose(I, (3, 2, 1, 0)).reshape(n, c, -1).astype(f32))     # column-major h,w,c,n
    np.testing.assert_array_equal(dh.mask, _cm(mask).astype(f32))
    np.testing.assert_array_equal(dh.K, K.T.reshape(-1).astype(f32))                                      # K[0]=fx K[4]=fy K[6]=cx K[7]=cy
    np.testing.assert_array_equal(dh.z0, np.transpose(z0, (2, 1, 0)).reshape(2, -1).astype(f32))
    with pytest.raises(RuntimeError, match="not found"):
        scipy.io.savemat(path, {"I": I, "K": K})
        host.load_dataset("matlab", path, preprocess=False)
    with pytest.raises(RuntimeError, match="Failed opening"):
        host.load_dataset("matlab", str(tmp_path / "missing.mat"))


def test_mat5_writer_is_read_by_scipy(host, tmp_path):
    import ctypes as C
    x = np.arange(17, dtype=f32) * 0.5
    p = str(tmp_path / "x.mat")
    assert host.load().srps_host_write_mat_floats(x.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(x.size), p.encode()) == 0
    m = scipy.io.loadmat(p)["x"]                       # variable "x", single, [len, 1]  (Utilities.cpp:46-63)
    assert m.dtype == np.float32 and m.shape == (17, 1)
    np.testing.assert_array_equal(m[:, 0], x)


def _make_folder(tmp_path, h=12, w=16, sf=2, n=11, rgba=False):
    from PIL import Image
    rng = np.random.default_rng(1)
    root = tmp_path / "ds"; (root / "RGB").mkdir(parents=True); (root / "Depth").mkdir()
    imgs = {}
    for i in range(1, n + 1):
        a = rng.integers(0, 256, size=(h, w, 4 if rgba else 3), dtype=np.uint8)
        Image.fromarray(a, "RGBA" if rgba else "RGB").save(root / "RGB" / f"I_{i}.png")
        imgs[f"I_{i}.png"] = a[:, :, :3]
    mask = (rng.uniform(size=(h, w)) > 0.3).astype(np.uint8) * 255
    Image.fromarray(mask, "L").save(root / "mask.png")
    depth = rng.integers(0, 65536, size=(2, h // sf, w // sf)).astype(np.uint16)
    for k in range(2):
        Image.fromarray(depth[k]).save(root / "Depth" / f"D_{k}.png")        # mode I;16
    (root / "K.txt").write_text("1216.73000000000,0,639.500000000000\n0,1216.73000000000,479.500000000000\n0,0,1\n%d,0,9870" % sf)
    return root, imgs, mask, depth


@pytest.mark.parametrize("rgba", [False, True])
def test_image_folder_loader(host, tmp_path, rgba):
    root, imgs, mask, depth = _make_folder(tmp_path, rgba=rgba)
    dh = host.load_dataset("images", str(root), preprocess=False)
    h, w = mask.shape
    assert (dh.I_h, dh.I_w, dh.I_c, dh.I_n, dh.sf, dh.z0_n) == (h, w, 3, 11, 2, 2)
    order = sorted(imgs)                                           # cv::glob order: I_1, I_10, I_11, I_2, ...
    assert order[:4] == ["I_1.png", "I_10.png", "I_11.png", "I_2.png"]
    for n, name in enumerate(order):
        for c in range(3):                                         # plane 0 = R, 1 = G, 2 = B, values /255
            np.testing.assert_array_equal(dh.I[n, c], _cm(imgs[name][:, :, c]).astype(f32) / f32(255))
    np.testing.assert_array_equal(dh.mask, _cm(mask).astype(f32) / f32(255))
    np.testing.assert_allclose(dh.K, [1216.73, 0, 0, 0, 1216.73, 0, 639.5, 479.5, 1], rtol=1e-6)
    exp = f32(0) + (depth.astype(f32) / f32(65535)) * f32(9870)    # Utilities.cpp:330, 392
    for k in range(2):
        np.testing.assert_allclose(dh.z0[k], _cm(exp[k]), rtol=1e-6)


def test_png_decoder_on_the_reference_dataset(host):
    root = "/root/reference/dataset/Images/Mitten"
    if not os.path.isdir(root):
        pytest.skip("reference dataset not present on this machine")
    from PIL import Image
    dh = host.load_dataset("images", root, preprocess=False)
    assert (dh.I_h, dh.I_w, dh.I_n, dh.sf) == (960, 1280, 20, 2)
    a = np.asarray(Image.open(os.path.join(root, "RGB", "I_10.png")).convert("RGB"))     # second in glob order
    np.testing.assert_array_equal(dh.I[1, 0], _cm(a[:, :, 0]).astype(f32) / f32(255))
    m = np.asarray(Image.open(os.path.join(root, "mask.png")).convert("L"))
    np.testing.assert_array_equal(dh.mask, _cm(m).astype(f32) / f32(255))
    assert int(dh.mask.sum()) == 148600
    d = np.asarray(Image.open(os.path.join(root, "Depth", sorted(os.listdir(os.path.join(root, "Depth")))[0])))
    np.testing.assert_allclose(dh.z0[0], _cm(d.astype(f32) / f32(65535) * f32(9870)), rtol=1e-6)


# ---- pre-processing against independent restatements -------------------------------------------
def _cubic_w(x, A=-0.75):
    x = abs(x)
    if x <= 1: return (A + 2) * x ** 3 - (A + 3) * x ** 2 + 1
    if x < 2: return A * x ** 3 - 5 * A * x ** 2 + 8 * A * x - 4 * A
    return 0.0


def test_resize_cubic_is_the_keys_kernel_with_half_pixel_centres(host):
    rng = np.random.default_rng(2)
    src = rng.uniform(size=(7, 9)).astype(f32)
    out = host.resize_cubic(src, 14, 27)
    ref = np.zeros((14, 27))
    for i in range(14):
        fy = (i + 0.5) * 7 / 14 - 0.5; y0 = int(np.floor(fy))
        for j in range(27):
            fx = (j + 0.5) * 9 / 27 - 0.5; x0 = int(np.floor(fx))
            acc = 0.0
            for a in range(-1, 3):
                for b in range(-1, 3):
                    acc += _cubic_w(fy - (y0 + a)) * _cubic_w(fx - (x0 + b)) * src[min(max(y0 + a, 0), 6), min(max(x0 + b, 0), 8)]
            ref[i, j] = acc
    np.testing.assert_allclose(out, ref, atol=2e-6)
    np.testing.assert_allclose(host.resize_cubic(np.full((5, 5), 3.0, f32), 10, 10), 3.0, atol=1e-6)     # partition of unity


def test_bilateral_filter_definition(host):
    rng = np.random.default_rng(3)
    src = rng.uniform(size=(9, 11)).astype(f32)
    out = host.bilateral(src, 2.0, 2.0)
    pad = np.pad(src.astype(np.float64), 3, mode="reflect")          # BORDER_REFLECT_101
    ref = np.zeros_like(src, dtype=np.float64)
    for i in range(9):
        for j in range(11):
            s = ws = 0.0
            for di in range(-3, 4):
                for dj in range(-3, 4):
                    if di * di + dj * dj > 9: continue
                    v = pad[i + 3 + di, j + 3 + dj]
                    wgt = np.exp(-(di * di + dj * dj) / 8.0 - (v - src[i, j]) ** 2 / 8.0)
                    s += wgt * v; ws += wgt
            ref[i, j] = s / ws
    np.testing.assert_allclose(out, ref, atol=1e-5)


def test_inpainting_fills_holes_smoothly(host):
    ii, jj = np.meshgrid(np.arange(40), np.arange(48), indexing="ij")
    truth = (1.0 + 0.01 * ii + 0.02 * jj + 0.1 * np.sin(ii / 9.0)).astype(f32)
    flag = np.zeros_like(truth, dtype=np.uint8)
    flag[10:18, 20:30] = 1; flag[30, 5] = 1; flag[0:3, 0:4] = 1             # a block, a single pixel, a corner
    img = truth.copy(); img[flag == 1] = 0
    out = host.inpaint(img, flag, 16)
    np.testing.assert_array_equal(out[flag == 0], truth[flag == 0])          # known pixels untouched
    assert np.abs(out - truth)[flag == 1].max() < 0.06                        # ramp + gentle curvature recovered (image range 1..2.4)
    assert np.abs(out - truth)[10:18, 20:30].max() < 0.03
    assert out[flag == 1].min() >= truth.min() - 0.05 and out[flag == 1].max() <= truth.max() + 0.05
    np.testing.assert_array_equal(host.inpaint(truth, np.zeros_like(flag), 16), truth)      # nothing flagged -> identity


def test_preprocess_chain_on_clean_depth(host, oracle):
    """no zeros -> no inpainting; chain = channel mean (divide by nc) -> /max -> bilateral -> *max -> cubic x sf"""
    rng = np.random.default_rng(4)
    zh, zw, nc, sf = 10, 14, 2, 2
    z0 = rng.uniform(1.0, 1.2, size=(nc, zh * zw)).astype(f32)
    zs, zf = host.preprocess_depth(z0, zh, zw, nc, zh * sf, zw * sf)
    mean, flag = oracle.mean_across_channels(z0, zh, zw, nc)
    assert flag.sum() == 0
    img = mean.reshape(zw, zh)                                                # the reference's transposed view, SRPS.cu:130
    mx = img.max()
    sm = host.bilateral(img / mx, 2.0, 2.0) * mx
    np.testing.assert_allclose(zs.reshape(zw, zh), sm, rtol=1e-6)
    np.testing.assert_allclose(zf.reshape(zw * sf, zh * sf), host.resize_cubic(sm, zw * sf, zh * sf), rtol=1e-6)


def test_headless_views_replace_imshow(host, tmp_path):
    """normals / albedo / depth views of Utilities.cpp:242-320 written as PNG and read back with PIL"""
    from PIL import Image
    rng = np.random.default_rng(5)
    rows, cols = 12, 10
    m = rng.uniform(size=(rows, cols)) > 0.3
    imask = np.nonzero(_cm(m))[0].astype(np.int32)
    P = imask.size
    N = rng.normal(size=(4, P)).astype(f32); N[:3] /= np.linalg.norm(N[:3], axis=0); N[3] = 1
    host.write_view("normals", N, imask, rows, cols, str(tmp_path / "n.png"))
    img = np.asarray(Image.open(tmp_path / "n.png")).astype(np.float64) / 255
    ref = np.zeros((rows, cols, 3))
    rr, cc = imask % rows, imask // rows
    ref[rr, cc, 0] = np.clip(0.5 + 0.5 * N[0], 0, 1); ref[rr, cc, 1] = np.clip(0.5 + 0.5 * N[1], 0, 1); ref[rr, cc, 2] = np.clip(0.5 - 0.5 * N[2], 0, 1)
    ref = (ref - ref.min()) / (ref.max() - ref.min())
    assert img.shape == (rows, cols, 3) and np.abs(img - ref).max() <= 0.5 / 255 + 1e-6
    rho = rng.uniform(0.1, 0.9, size=(3, P)).astype(f32); rho[1, 0] = 50.0                      # an outlier is capped at median + 5 sigma
    host.write_view("albedo", rho, imask, rows, cols, str(tmp_path / "a.png"))
    img = np.asarray(Image.open(tmp_path / "a.png")).astype(np.float64) / 255
    cap = np.median(rho, axis=1) + 5 * rho.std(axis=1)
    ref = np.zeros((rows, cols, 3)); ref[rr, cc] = np.clip(np.minimum(rho, cap[:, None]), 0, 1).T
    assert np.abs(img - ref).max() <= 0.5 / 255 + 2e-3
    z = rng.uniform(1, 2, size=P).astype(f32)
    host.write_view("depth", z, imask, rows, cols, str(tmp_path / "z.png"), scale=0.5)
    img = np.asarray(Image.open(tmp_path / "z.png"))
    assert img.shape == (6, 5, 3) and img.max() > 100


def test_cli_help_and_errors(host, pkg):
    cli = pkg.host.CLI
    out = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--dstype" in out.stdout and "--dsloc" in out.stdout and "--blockx" in out.stdout
    out = subprocess.run([cli], capture_output=True, text=True)                # no --dsloc: prints help, returns 0 (Main.cpp:23-26)
    assert out.returncode == 0 and "Usage" in out.stdout
    out = subprocess.run([cli, "-d", "/nonexistent.mat"], capture_output=True, text=True)
    assert out.returncode == 1 and "Failed opening MAT file" in out.stderr
    # round 4: a --gpus / --device / --partition value that is no number (or no partition) ends in a message + usage + exit code 1
    out = subprocess.run([cli, "--gpus", "x", "-d", "/nonexistent.mat"], capture_output=True, text=True)
    assert out.returncode == 1 and "--gpus: 'x' is not a number" in out.stderr and "Usage" in out.stdout
    out = subprocess.run([cli, "--gpus=", "-d", "/nonexistent.mat"], capture_output=True, text=True)
    assert out.returncode == 1 and "is not a number" in out.stderr
    out = subprocess.run([cli, "--partition", "rows", "-d", "/nonexistent.mat"], capture_output=True, text=True)
    assert out.returncode == 1 and "--partition: images or strips" in out.stderr
    assert "--partition" in subprocess.run([cli, "--help"], capture_output=True, text=True).stdout
    # round 3: the multi-GPU switches are part of the command line; a bad data set fails before any device is touched
    out = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert "--gpus" in out.stdout and "--sharded" in out.stdout
    out = subprocess.run([cli, "--gpus", "4", "-d", "/nonexistent.mat"], capture_output=True, text=True)
    assert out.returncode == 1 and "Failed opening MAT file" in out.stderr

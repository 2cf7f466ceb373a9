"""TEST INFRASTRUCTURE ONLY.  Expected outputs of the input pipeline's image transforms (vfmseg_amd/datasets.py: the mmseg transform chain
of configs/_base_/datasets/*.py, whose OpenCV kernels are not available offline) computed by INDEPENDENT implementations that share no code
with the product: a float64 loop-free numpy bilinear / nearest resize written from the sampling formula (OpenCV INTER_LINEAR /
INTER_NEAREST: half-pixel centres, edge clamp; floor(dst * scale)), Python's `colorsys` for the 8-bit HSV convention (H in [0, 180)), and
numpy indexing for flips.  Writes tests/golden/transforms.npz (inputs + expected outputs).

    python -m oracle.gen_transform_fixture"""
import colorsys
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def bilinear_u8(img, oh, ow):
    h, w = img.shape[:2]
    ys = np.clip((np.arange(oh) + 0.5) * (h / oh) - 0.5, 0, h - 1)
    xs = np.clip((np.arange(ow) + 0.5) * (w / ow) - 0.5, 0, w - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    f = img.astype(np.float64).reshape(h, w, -1)
    top = f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx
    bot = f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx
    return (top * (1 - fy) + bot * fy).reshape((oh, ow) + img.shape[2:])   # float64, un-rounded: the test allows the rounding tie


def nearest(lab, oh, ow):
    h, w = lab.shape
    ys = np.minimum(np.floor(np.arange(oh) * (h / oh)).astype(int), h - 1)
    xs = np.minimum(np.floor(np.arange(ow) * (w / ow)).astype(int), w - 1)
    return lab[ys][:, xs]


def bgr2hsv_ref(img):
    out = np.zeros(img.shape, np.float64)
    for i in range(img.shape[0]):
        for j in range(img.shape[1]):
            b, g, r = (img[i, j].astype(np.float64) / 255.0)
            hh, ss, vv = colorsys.rgb_to_hsv(r, g, b)
            out[i, j] = (hh * 180.0, ss * 255.0, vv * 255.0)
    return out   # un-rounded


def hsv2bgr_ref(hsv):
    out = np.zeros(hsv.shape, np.float64)
    for i in range(hsv.shape[0]):
        for j in range(hsv.shape[1]):
            hh, ss, vv = hsv[i, j].astype(np.float64)
            r, g, b = colorsys.hsv_to_rgb((hh * 2.0 / 360.0) % 1.0, ss / 255.0, vv / 255.0)
            out[i, j] = (b * 255.0, g * 255.0, r * 255.0)
    return out


def main():
    rng = np.random.RandomState(20240)
    h, w = 24, 40
    base = rng.randint(0, 256, (6, 10, 3))
    img = np.clip(np.kron(base, np.ones((4, 4, 1))) + rng.randint(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)   # smooth blocks + noise
    img[0, 0] = (0, 0, 0)
    img[0, 1] = (255, 255, 255)
    img[0, 2] = (128, 128, 128)      # grey: saturation 0, hue undefined -> 0
    img[0, 3] = (0, 0, 255)          # pure red in BGR
    lab = np.kron(rng.randint(0, 19, (6, 10)), np.ones((4, 4))).astype(np.uint8)
    lab[:3] = 255
    out = dict(img=img, lab=lab)
    for name, (oh, ow) in dict(up=(36, 60), down=(15, 25), odd=(31, 17)).items():
        out[f"resize_{name}_size"] = np.array([oh, ow])
        out[f"resize_{name}_img"] = bilinear_u8(img, oh, ow)
        out[f"resize_{name}_lab"] = nearest(lab, oh, ow)
    hsv = bgr2hsv_ref(img)
    out["hsv"] = hsv
    hsv_u8 = np.stack([np.rint(hsv[..., 0]) % 180, np.rint(hsv[..., 1]), np.rint(hsv[..., 2])], -1).astype(np.uint8)
    out["hsv_u8"] = hsv_u8
    out["bgr_back"] = hsv2bgr_ref(hsv_u8)
    out["flip_h_img"], out["flip_h_lab"] = img[:, ::-1].copy(), lab[:, ::-1].copy()
    out["flip_v_img"], out["flip_v_lab"] = img[::-1].copy(), lab[::-1].copy()
    np.savez_compressed(os.path.join(GOLD, "transforms.npz"), **out)
    print("wrote", os.path.join(GOLD, "transforms.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

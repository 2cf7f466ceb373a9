"""python tools/convert_models.py {dinov2,eva02,sam,clip} PRETRAINED CONVERTED [--kernel 16 --height 512 --width 512]
(the four scripts of the reference's tools/convert_models/ behind one entry point; same arguments)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfmseg_amd import convert  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("family", choices=["dinov2", "eva02", "sam", "clip"])
    ap.add_argument("pretrained")
    ap.add_argument("converted")
    ap.add_argument("--kernel", default=16, type=int)
    ap.add_argument("--height", default=512, type=int)
    ap.add_argument("--width", default=512, type=int)
    ap.add_argument("--embed_dim", default=1024, type=int)
    a = ap.parse_args()
    if a.family == "clip":
        sd = torch.jit.load(a.pretrained, map_location="cpu").float().state_dict()
        out = convert.convert_clip(sd, a.height, a.kernel, a.embed_dim)
    else:
        sd = torch.load(a.pretrained, map_location="cpu")
        if len(sd.keys()) <= 10 and "model" not in sd:
            raise KeyError(f"the read weights may be abnormal: {list(sd.keys())}")
        if a.family == "dinov2":
            out = convert.convert_dinov2(sd, a.kernel, (a.height, a.width))
        elif a.family == "eva02":
            out = convert.convert_eva02(sd, a.kernel, a.height // a.kernel)
        else:
            out = convert.convert_sam(sd, a.kernel, (a.height, a.width))
    torch.save(out, a.converted)
    print("saved", a.converted)


if __name__ == "__main__":
    main()

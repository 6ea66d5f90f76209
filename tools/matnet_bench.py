"""MaterialNet (DINOv2 ViT-B/14 + two DPT heads, random name-seeded weights) inference time on one MI355X: BASELINE configs[3]
shape (1024x1024 input, network input 518 -> 1022 on the long side as `infer_image` resizes it).  usage: python tools/matnet_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialist_amd.materialnet import MaterialNet, init_from_names, network_input_size  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    net = init_from_names(MaterialNet()).to(dev).eval()
    nparam = sum(p.numel() for p in net.parameters())
    img = (np.random.default_rng(0).random((1024, 1024, 3)) * 255).astype(np.uint8)
    for input_size in (518, 1022):
        nw, nh = network_input_size(1024, 1024, input_size)
        x = torch.rand(1, 3, nh, nw, device=dev)
        for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
            with torch.no_grad(), ctx:
                for _ in range(3):
                    net(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    net(x)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"MaterialNet forward, network input {nh}x{nw} ({(nh // 14) * (nw // 14)} tokens), {name}: {ms:.1f} ms  ({nparam / 1e6:.1f} M parameters)")
    with torch.no_grad():
        net.infer_image(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.infer_image(img)
        torch.cuda.synchronize()
        print(f"infer_image(1024x1024 uint8 -> 5 maps on the host, fp32): {(time.perf_counter() - t0) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()

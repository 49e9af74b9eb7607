"""End-to-end rate of BASELINE.json configs[4] on one MI355X: a DINOv2-ViT-L/14-shaped transformer (random weights: no
network for checkpoints) -> forward hooks -> HBM reservoir -> configs[1]-shaped SAE train step, no disk in between.

    python tools/bench_extract_e2e.py [--images 4096] [--img-batch 64] [--autocast bf16|none]

Prints activations/s of the whole pipeline and the split between the transformer forward and the SAE steps."""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from saev_amd import data, nn  # noqa: E402
from saev_amd.data.vit import VisionTransformer  # noqa: E402
from saev_amd.framework import train as T  # noqa: E402
from saev_amd.nn import objectives  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--img-batch", type=int, default=64)
    ap.add_argument("--autocast", default="bf16")
    ap.add_argument("--layer", type=int, default=23)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    vit = VisionTransformer.vit_l14().to(dev).eval()
    rec = data.ActivationRecorder(vit, vit.blocks, layers=(args.layer,), content_tokens_per_example=256, cls_token=True)
    imgs = torch.randn(args.img_batch, 3, 224, 224, device=dev)  # one synthetic image batch, reused (the decode/augment side is not ours)
    t_fwd = [0.0]

    class Timed(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if args.autocast == "bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = self.inner(x)
            else:
                y = self.inner(x)
            torch.cuda.synchronize()
            t_fwd[0] += time.perf_counter() - t0
            return y

    rec.model = Timed(vit)

    def images():
        for lo in range(0, args.images, args.img_batch):
            yield imgs, torch.arange(lo, lo + args.img_batch)

    B = 16384
    feed = data.ExtractionFeed(data.ExtractConfig(layer=args.layer, batch_size=B, buffer_size=8, seed=1), rec, images,
                               n_examples=args.images, d_model=1024, device=dev)
    cfg = T.Config(n_train=args.images * 256, sae=nn.SparseAutoencoderConfig(d_model=1024, d_sae=32768, reinit_blend=0.0),
                   objective=objectives.Matryoshka(n_prefixes=1), log_every=10**9, track=False,
                   train_data=data.ShuffledConfig(batch_size=B), val_data=data.ShuffledConfig(batch_size=B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    saes, objs, run, steps = T.train([cfg], train_feed=feed)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = args.images * 256
    print(f"{steps} SAE steps on {n} activations in {dt:.2f} s = {n / dt / 1e6:.3f} M activations/s end to end; "
          f"transformer forward {t_fwd[0]:.2f} s ({100 * t_fwd[0] / dt:.0f} %, {args.autocast}), "
          f"everything else (hand-off + SAE steps + init) {dt - t_fwd[0]:.2f} s = {(dt - t_fwd[0]) / steps * 1e3:.2f} ms per step")


if __name__ == "__main__":
    main()

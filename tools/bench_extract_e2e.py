"""End-to-end rate of BASELINE.json configs[4] on one MI355X: a DINOv2-ViT-L/14-shaped transformer (random weights: no
network for checkpoints) -> forward hooks -> HBM reservoir -> configs[1]-shaped SAE train step, no disk in between.

    python tools/bench_extract_e2e.py [--images 4096] [--img-batch 64] [--autocast bf16|none]

Prints activations/s of the whole pipeline and the split between the transformer forward and the SAE steps; `run()` is also
what `bench.py --extract-e2e` puts on its line as the `extract_e2e` sub-record."""
import argparse
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from saev_amd import data, nn  # noqa: E402
from saev_amd.data.vit import VisionTransformer  # noqa: E402
from saev_amd.framework import train as T  # noqa: E402
from saev_amd.nn import objectives  # noqa: E402


def run(dev, images: int = 4096, img_batch: int = 64, autocast: str = "bf16", layer: int = 23) -> dict:
    args = argparse.Namespace(images=images, img_batch=img_batch, autocast=autocast, layer=layer)
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
    return {"workload": "configs[4] on one GPU: ViT-L/14-shaped transformer (24 blocks, d_model 1024, 257 tokens, random init, "
                        f"{args.autocast} forward, stock PyTorch-ROCm) -> hooks on block {args.layer} -> device reservoir -> "
                        "d_sae=32768 k=32 SAE train steps of 16384 rows, no disk in between",
            "images": args.images, "image_batch": args.img_batch, "activations": n, "sae_steps": steps, "seconds": dt,
            "activations_per_sec_end_to_end": n / dt, "transformer_forward_seconds": t_fwd[0],
            "transformer_forward_share": t_fwd[0] / dt,
            "handoff_plus_sae_ms_per_step": (dt - t_fwd[0]) / steps * 1e3,
            "note": "the transformer forward is PyTorch's, not this package's; what is ours is everything after the hooks"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--img-batch", type=int, default=64)
    ap.add_argument("--autocast", default="bf16")
    ap.add_argument("--layer", type=int, default=23)
    args = ap.parse_args()
    r = run(torch.device("cuda", 0), args.images, args.img_batch, args.autocast, args.layer)
    print(f"{r['sae_steps']} SAE steps on {r['activations']} activations in {r['seconds']:.2f} s = "
          f"{r['activations_per_sec_end_to_end'] / 1e6:.3f} M activations/s end to end; transformer forward "
          f"{r['transformer_forward_seconds']:.2f} s ({100 * r['transformer_forward_share']:.0f} %, {args.autocast}), everything else "
          f"(hand-off + SAE steps + init) = {r['handoff_plus_sae_ms_per_step']:.2f} ms per step")


if __name__ == "__main__":
    main()

"""HBM footprint of one SAE engine per BASELINE configuration: the four flat buffers the host owns
plus the context's own scratch, as the library reports it (saev_scratch_bytes).  Needs a GPU (the engine allocates for real).

    python tools/footprint.py > gpurun_out/r04_footprint.txt"""
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

CONFIGS = [
    ("configs[0] d=768 S=6144 k=32 B=4096 f16r", dict(d_model=768, d_sae=6144, top_k=32, max_batch=4096)),
    ("configs[1]/[2] d=1024 S=32768 k=32 B=16384 f16r", dict(d_model=1024, d_sae=32768, top_k=32, max_batch=16384)),
    ("  same, P=10 Matryoshka blocks", dict(d_model=1024, d_sae=32768, top_k=32, max_batch=16384, _prefixes=10)),
    ("  same, aux_dead_cap = d_sae (the old default)", dict(d_model=1024, d_sae=32768, top_k=32, max_batch=16384, aux_dead_cap=32768)),
    ("  same, sparse exchange at 8 ranks x 2048 rows", dict(d_model=1024, d_sae=32768, top_k=32, max_batch=2048, max_backward_rows=16384)),
    ("configs[3] d=1280 S=81920 k=64 B=16384 bf16", dict(d_model=1280, d_sae=81920, top_k=64, max_batch=16384, encoder="bf16")),
]
GB = 1e9
print(f"{'configuration':52s} {'flat x4':>9s} {'scratch':>9s} {'AuxK':>9s} {'Matry.':>9s} {'total GB':>9s}")
for name, kw in CONFIGS:
    P = kw.pop("_prefixes", 0)
    eng = SaeEngine(EngineConfig(**kw), "cuda:0")
    if P:
        eng.set_prefixes([(i + 1) * kw["d_sae"] // P for i in range(P)])
    flat = 4 * 4 * eng.n_params
    allb, aux, mat = eng.scratch_bytes(0), eng.scratch_bytes(1), eng.scratch_bytes(2)
    print(f"{name:52s} {flat / GB:9.2f} {(allb - aux - mat) / GB:9.2f} {aux / GB:9.2f} {mat / GB:9.2f} {(flat + allb) / GB:9.2f}")
    eng.close()
    del eng
    torch.cuda.empty_cache()

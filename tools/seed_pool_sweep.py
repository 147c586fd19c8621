"""How many sample tiles should a corpus shard contribute to the pooled thresholds of an 8-shard join?  100 k x 8 x 125 k x 768
fp16, k = 10, on one GPU (TUNING build: LVS_TILE_SEED_DIV_BIG sets the tiles per shard).  Development aid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend

be = HipBackend("cuda:0")
n, nq, d, k, W = 1_000_000, 100_000, 768, 10, 8
g = torch.Generator(device=be.device); g.manual_seed(1)
xb = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=be.device), dim=1).to(torch.float16)
j = torch.randint(0, n, (nq,), generator=g, device=be.device)
xq = torch.nn.functional.normalize(0.7 * xb[j].float() + 0.7 * torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=be.device), dim=1), dim=1).to(torch.float16)
cb, cq = be.pack(xb, _capi.PACK_F16), be.pack(xq, _capi.PACK_F16)
del xb, xq
per = n // W
shards = [be.slice_rows(cb, r * per, (r + 1) * per) for r in range(W)]
ref = be.search_keys(cb, cq, k, 0)
for div in (512, 24, 12, 8):
    os.environ["LVS_TILE_SEED_DIV_BIG"] = str(div)
    tiles = be.seed_tiles(nq, per, k, 0, 0)

    def pooled():
        return torch.cat([be.seed_scores(sh, cq, 0, tiles) for sh in shards])

    def run(seeds):
        return torch.stack([be.search_keys(sh, cq, k, 0, id_offset=r * per, seed_scores=seeds) for r, sh in enumerate(shards)])

    run(pooled()); be.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); seeds = pooled(); e1.record()
    be.timing_enable(True)
    parts = run(seeds); be.synchronize()
    tot, cnt = be.timing_read(); be.timing_enable(False)
    same = bool(torch.equal(be.merge_keys(parts), ref))
    print(f"{tiles:3d} tiles per shard ({W * tiles} pooled): sample pass {e0.elapsed_time(e1) / W:5.2f} ms + search {tot / cnt:6.2f} ms per shard = "
          f"{e0.elapsed_time(e1) / W + tot / cnt:6.2f} ms   merged == single launch: {same}", flush=True)

"""Timing ablations of lvs_ra_kernel (TUNING build; LVS_RA_DEBUG variants give WRONG results): kernel ms at nq points x 1 024 x 768.
usage: python tools/ra_ablate.py [nq] [debug values ...]
(needs the experiment kernel of tools/lvs_ra_experiment.hip.txt built into the TUNING library - the shipped sources do not contain it)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend, _ptr
be = HipBackend("cuda:0")
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dbgs = sys.argv[2:] or ["0", "1", "2", "3", "7", "15"]
K, d = 1024, 768
g = torch.Generator(device=be.device); g.manual_seed(3)
cen = torch.randn((K, d), generator=g, device=be.device) * 0.5
x = (cen[torch.randint(0, K, (nq,), generator=g, device=be.device)] + 0.6 * torch.randn((nq, d), generator=g, device=be.device)).to(torch.float16)
pq, pc = be.pack(x, _capi.PACK_F16), be.pack(cen, _capi.PACK_SPLIT)
del x
ws = torch.empty((int(be.lib.lvs_nearest3_workspace_bytes(nq, K, d)),), dtype=torch.uint8, device=be.device)
keys = torch.zeros((nq,), dtype=torch.int64, device=be.device); keys2 = torch.zeros_like(keys)
sec = torch.zeros((nq,), dtype=torch.float32, device=be.device); third = torch.zeros_like(sec)
def call():
    be._c("lvs_nearest3", _ptr(pc.rows), pc.mode, pc.n, _ptr(pq.rows), pq.mode, nq, d, 1, _ptr(pc.norms), _ptr(pq.norms), 0,
          _ptr(keys), _ptr(keys2), _ptr(sec), _ptr(third), _ptr(ws), int(ws.numel()), be._stream())
for rnd in range(2):
    for tag, env in [("lvs_assign_kernel", {"LVS_RA": "0"})] + [(f"lvs_ra_kernel DBG={v}", {"LVS_RA": "1", "LVS_RA_DEBUG": v}) for v in dbgs]:
        for k in ("LVS_RA", "LVS_RA_DEBUG", "LVS_RA_IPW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        if os.environ.get("RA_IPW"):
            os.environ["LVS_RA_IPW"] = os.environ["RA_IPW"]
        call(); be.synchronize()
        be.timing_enable(True)
        for _ in range(3):
            call()
        be.synchronize()
        tot, cnt = be.timing_read(); be.timing_enable(False)
        ms = tot / max(cnt, 1)
        print(f"{tag:28s} {ms:8.3f} ms  {2 * nq * K * d / ms / 1e9 / 2500:6.3f} of the roof", flush=True)

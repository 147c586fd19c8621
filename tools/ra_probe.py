"""lvs_ra_kernel (points resident) against lvs_assign_kernel through lvs_nearest3, TUNING build (LVS_RA = 0 / 1): outputs compared
bit for bit (best key, second key, second score, third score), kernel time from the library's HIP events (development aid).
usage: python tools/ra_probe.py [nq ...]   env: RA_K (1024), RA_D (768), RA_METRIC (1 = L2), RA_SPLIT (0)
(needs the experiment kernel of tools/lvs_ra_experiment.hip.txt built into the TUNING library - the shipped sources do not contain it)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lotus_amd import _capi
_capi.load(os.path.join(ROOT, "lotus_amd", "liblotus_hip_tuning.so"))
from lotus_amd.backend import HipBackend, _ptr

be = HipBackend("cuda:0")
K, d, metric = int(os.environ.get("RA_K", 1024)), int(os.environ.get("RA_D", 768)), int(os.environ.get("RA_METRIC", 1))
split = int(os.environ.get("RA_SPLIT", 0))
sizes = [int(v) for v in sys.argv[1:]] or [100_000]
g = torch.Generator(device=be.device); g.manual_seed(3)
for nq in sizes:
    cen = torch.randn((K, d), generator=g, device=be.device) * 0.5
    lab = torch.randint(0, K, (nq,), generator=g, device=be.device)
    x = cen[lab] + 0.6 * torch.randn((nq, d), generator=g, device=be.device)
    pq = be.pack(x.to(torch.float16) if not split else x, _capi.PACK_F16 if not split else _capi.PACK_SPLIT)
    pc = be.pack(cen, _capi.PACK_SPLIT)
    del x
    need = int(be.lib.lvs_nearest3_workspace_bytes(nq, K, d))
    ws = torch.empty((need,), dtype=torch.uint8, device=be.device)
    outs = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["LVS_RA"] = mode
        keys = torch.zeros((nq,), dtype=torch.int64, device=be.device); keys2 = torch.zeros_like(keys)
        sec = torch.zeros((nq,), dtype=torch.float32, device=be.device); third = torch.zeros_like(sec)
        def call():
            be._c("lvs_nearest3", _ptr(pc.rows), pc.mode, pc.n, _ptr(pq.rows), pq.mode, nq, d, metric, _ptr(pc.norms), _ptr(pq.norms), 0,
                  _ptr(keys), _ptr(keys2), _ptr(sec), _ptr(third), _ptr(ws), int(ws.numel()), be._stream())
        call(); be.synchronize()
        be.timing_enable(True)
        for _ in range(3):
            call()
        be.synchronize()
        tot, cnt = be.timing_read(); be.timing_enable(False)
        res = (keys.cpu().numpy().copy(), keys2.cpu().numpy().copy(), sec.cpu().numpy().view(np.uint32).copy(), third.cpu().numpy().view(np.uint32).copy())
        line = f"nq {nq} K {K} d {d} metric {metric} split {split} LVS_RA={mode}: kernel {tot / max(cnt, 1):8.3f} ms"
        if mode in outs:
            pass
        if "0" in outs and mode == "1":
            ref = outs["0"]
            eq = [bool(np.array_equal(a, b)) for a, b in zip(ref, res)]
            line += f"  equal to lvs_assign_kernel (keys, keys2, second, third): {eq}"
            if not all(eq):
                bad = np.nonzero(ref[0] != res[0])[0]
                line += f"  first differing rows {bad[:8].tolist()} of {bad.size}"
        outs.setdefault(mode, res)
        print(line, flush=True)

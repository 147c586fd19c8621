"""Reference point: vendor GEMM (hipBLASLt through torch.mm) on the bench's operand shapes, no top-k."""
import time, torch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
d = 768
for nq, nb in ((100000, 65536), (65536, 65536), (16384, 131072)):
    a = torch.nn.functional.normalize(torch.randn((nq, d), generator=g, device=dev), dim=1).half()
    b = torch.nn.functional.normalize(torch.randn((nb, d), generator=g, device=dev), dim=1).half()
    for _ in range(2): c = a @ b.t()
    torch.cuda.synchronize(); t0 = time.time(); n = 5
    for _ in range(n): c = a @ b.t()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"torch.mm {nq}x{nb}x{d} fp16: {dt*1e3:.2f} ms  {2.0*nq*nb*d/dt/1e12:.0f} TFLOP/s (writes {nq*nb*2/1e9:.1f} GB of scores)", flush=True)
    del c

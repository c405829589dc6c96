import os, sys, torch, time
sys.path.insert(0, os.getcwd())
import triple_hybrid_rag_amd as T
nd, nq = 1_000_000, 1024
parts = []
for i in range(10):   # 100K docs at a time: 6.5 GB of float32 noise each
    parts.append(torch.nn.functional.normalize(torch.randn(nd // 10, 128, 128, device="cuda"), dim=2).to(torch.float16))
rows = torch.cat(parts); del parts
print("token store GB", rows.numel() * 2 / 1e9, flush=True)
packed = T._native.maxsim_pack(rows)
qtok = torch.nn.functional.normalize(torch.randn(nq, 32, 128, device="cuda"), dim=2).to(torch.float16)
cand = torch.randint(0, nd, (nq, 100), device="cuda", dtype=torch.int32)
cand[:, 0] = nd - 1 - torch.arange(nq, device="cuda", dtype=torch.int32)   # the far end of the store
a = T._native.maxsim(qtok, rows, cand)
b = T._native.maxsim(qtok, packed, cand, packed=True)
torch.cuda.synchronize()
print("packed == row-major:", bool(torch.equal(a, b)), "finite:", bool(torch.isfinite(a).all()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): T._native.maxsim(qtok, packed, cand, packed=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("ms", round(ms, 3), "TB/s", round(nq * 100 * 32768 / ms / 1e9, 2))

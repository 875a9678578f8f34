"""Ceiling probe for the int8-slice Gram: what does the vendor int8 GEMM (hipBLASLt behind torch._int_mm) reach on the
shape counts[B x N] @ Zslices[N x (pairs * slices)]?  Experiment only (torch is not part of the product)."""
import json, time, torch
dev = torch.device("cuda:0")
out = []
for (M, K, N) in [(5120, 10048, 13328), (5120, 10048, 15232), (4096, 8192, 8192), (8192, 8192, 8192)]:
    a = torch.randint(0, 6, (M, K), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 127, (K, N), dtype=torch.int8, device=dev)
    bt = torch.randint(-128, 127, (N, K), dtype=torch.int8, device=dev).t()
    for name, bb in (("row", b), ("colmajor", bt)):
        try:
            for _ in range(3): c = torch._int_mm(a, bb)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): c = torch._int_mm(a, bb)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
            out.append({"M": M, "K": K, "N": N, "b": name, "ms": round(dt * 1e3, 3), "TOPS": round(2 * M * K * N / dt / 1e12, 1)})
        except Exception as e:
            out.append({"M": M, "K": K, "N": N, "b": name, "error": str(e)[:200]})
    del a, b, bt
for r in out: print(json.dumps(r))

import torch, json
dev = torch.device("cuda", 0)
n = 256 * 1000 * 1000
B = torch.rand(n, dtype=torch.float64, device=dev); C = torch.empty_like(B)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
ms = t(lambda: C.copy_(B)); print(json.dumps(dict(op="copy 2GB->2GB", ms=ms, tbs=2 * n * 8 / ms / 1e9)))
ms = t(lambda: C.fill_(1.0)); print(json.dumps(dict(op="fill 2GB", ms=ms, tbs=n * 8 / ms / 1e9)))
ms = t(lambda: B.sum()); print(json.dumps(dict(op="sum 2GB", ms=ms, tbs=n * 8 / ms / 1e9)))
ms = t(lambda: torch.add(B, 1.0, out=C)); print(json.dumps(dict(op="add 2GB->2GB", ms=ms, tbs=2 * n * 8 / ms / 1e9)))
ms = t(lambda: C.add_(B)); print(json.dumps(dict(op="C+=B (2 reads 1 write)", ms=ms, tbs=3 * n * 8 / ms / 1e9)))

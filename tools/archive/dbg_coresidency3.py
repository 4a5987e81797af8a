"""Shape of the damage: which elements of the spectrum of x come out wrong beside the matrix-core attention kernel?"""
import sys
import torch
sys.path.insert(0, ".")
from applecider_amd import hipops as H
dev = torch.device("cuda:0")
H.set_math("bf16x3")
g = torch.Generator().manual_seed(0)
B, L, Cin = 64, 1024, 64
x = torch.randn(B, L, Cin, generator=g).to(dev)
qkv = torch.randn(512, 129, 384, device=dev); pad = torch.zeros(512, 129, dtype=torch.uint8, device=dev)
size = (9, 1)
ref = H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, size)
torch.cuda.synchronize()
side = torch.cuda.Stream()
shown = 0
for it in range(60):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.no_grad():
            for _ in range(6):
                H.mha(qkv, pad, 8, 0.0, False)
    out = H.fft_rows_fwd(x, None, 0, L * Cin, Cin, 0, B, L, Cin, 0, size)
    torch.cuda.synchronize()
    bad = (out != ref)
    if bad.any():
        F, Bb, C2 = out.shape
        idx = bad.nonzero()
        fs, bs, cs = idx[:, 0], idx[:, 1], idx[:, 2]
        groups = sorted(set((int(b), int(c) // 32) for b, c in zip(bs.tolist(), cs.tolist())))
        print(f"run {it}: {int(bad.sum())} wrong of {out.numel()}; (batch row, 16-channel group) workgroups hit: {groups[:8]} ({len(groups)} in all); "
              f"frequencies hit: {int(fs.min())}..{int(fs.max())} ({len(set(fs.tolist()))} distinct); channel pairs in group: {sorted(set((int(c) % 32) // 4 for c in cs.tolist()))}", flush=True)
        f0 = int(fs[0]); b0 = int(bs[0]); c0 = int(cs[0]) // 4 * 4
        print("   first wrong entry (f, row, col0):", f0, b0, c0, "got", out[f0, b0, c0:c0 + 4].tolist(), "want", ref[f0, b0, c0:c0 + 4].tolist())
        col = out[:, b0, c0:c0 + 4]; rcol = ref[:, b0, c0:c0 + 4]
        wrongf = sorted(set(fs[(bs == b0)].tolist()))
        # is the wrong value the reference value of another frequency of the same column?
        hits = []
        for f in wrongf[:6]:
            d = (rcol - col[f]).abs().sum(1)
            j = int(d.argmin()); hits.append((f, j, float(d[j])))
        print("   wrong f -> nearest reference f' (L1 distance):", hits)
        print("   wrong frequencies:", wrongf)
        shown += 1
        if shown >= 5:
            break
print("done", shown)

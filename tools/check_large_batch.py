import sys, torch
sys.path.insert(0, '/root/repo')
from molann_amd import workloads as wl
dev = torch.device('cuda:0')
w = wl.get_workload('C3')
model = wl.build_model(w, dev).requires_grad_(False)
n = 40_000_003   # > 2^31 bytes of input (10.6 GB), ragged last tile
x = w.make_frames(1 << 20, device=dev, seed=3)
big = x.repeat(39, 1, 1)[:n - (1 << 20)]
big = torch.cat([big, x], 0)[:n].contiguous()
print(big.shape, big.numel() * 4 / 2**30, 'GiB')
with torch.no_grad():
    y = model(big)
    ref = model(x)
torch.cuda.synchronize()
ok = True
for blk in (0, 7, 19, 36):                       # whole copies of x inside the repeated part
    e = torch.equal(y[blk << 20:(blk + 1) << 20], ref)
    print('block', blk, e)
    ok &= e
for a, b in ((n - 5000, n), (38_951_000, 38_952_000), (2**31 // 264 - 500, 2**31 // 264 + 500)):
    with torch.no_grad():
        e = torch.equal(y[a:b], model(big[a:b].clone()))
    print('rows', a, b, e)
    ok &= e
print('large-batch rows equal small-batch rows:', ok)
assert ok

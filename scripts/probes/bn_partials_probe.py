"""Which element differs between the values a GEMM kernel takes its BatchNorm partials from and the values it
stores?  (tiny map, large mean: a 1-ulp difference in one element shows as 1e-3 of the variance)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from smaat_unet_amd import _lib
L = _lib.get(); dev = torch.device("cuda:0")
P = lambda t: None if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
N, C, M, H, W = 1, 16, 64, 2, 2
g = torch.Generator().manual_seed(3)
x = (torch.rand(N, C, H, W, generator=g) * 1e-2 + 5.0).to(dev)
w = (torch.rand(M, C, generator=g) * 0.5 + 0.75).to(dev)
wt = w.t().contiguous()
z = torch.empty(N, M, H, W, device=dev)
slots = L.smaat_pw_num_slots(N, H, W, M)
part = torch.full((3, slots, M), float("nan"), device=dev)
assert L.smaat_pointwise_fwd(P(x), C*H*W, P(wt), None, P(z), M*H*W, P(part), N, C, M, H, W, st) == 0
torch.cuda.synchronize()
print("slots", slots, "counts", part[2].sum(0)[:4].tolist())
zz = z.double().flatten(2)[0]            # [M][4]
mean = zz.mean(1); m2 = ((zz - mean[:, None])**2).sum(1)
pm = (part[2]*part[0]).double().sum(0)/part[2].double().sum(0)
print("mean diff (ulps of 80):", ((pm-mean)/7.63e-6)[:8].tolist())
pq = part[1].double().sum(0)
print("M2 kernel", pq[:6].tolist()); print("M2 ref   ", m2[:6].tolist())
ref = torch.einsum("mc,cp->mp", w.double(), x.double().flatten(2)[0])
print("z vs fp64 GEMM (ulps):", ((zz-ref)/7.63e-6)[:3].tolist())
# the same through the split kernel
Cp = (C + 15) // 16 * 16
pl = torch.empty((3, M, Cp), dtype=torch.int16, device=dev)
assert L.smaat_split_planes(P(w), M, C, P(pl), st) == 0
z2 = torch.empty(N, M, H, W, device=dev)
part2 = torch.full((3, L.smaat_pw_split_num_slots(N, H, W), M), float("nan"), device=dev)
assert L.smaat_pointwise_fwd_split(P(x), C*H*W, P(pl), None, P(z2), M*H*W, P(part2), N, C, M, H, W, st) == 0
torch.cuda.synchronize()
zz2 = z2.double().flatten(2)[0]
m2b = ((zz2 - zz2.mean(1)[:, None])**2).sum(1)
print("split: M2 kernel", part2[1].double().sum(0)[:6].tolist()); print("split: M2 ref   ", m2b[:6].tolist())
print("f32 vs split stored z, max ulps:", ((zz - zz2).abs().max() / 7.63e-6).item())

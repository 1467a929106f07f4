import sys, math, random, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import make_leaf
from oracle import leaf_oracle as lo
from leaf_pytorch_amd import _native
DEV = 'cuda:0'
SEEDS = [int(a) for a in sys.argv[1:]]
for seed in SEEDS:
  print('seed', seed, flush=True)
  rng = random.Random(7000 + seed)
  gen = torch.Generator().manual_seed(900 + seed)
  stream = _native.ALGO_FFT_WG | _native.ALGO_STREAM_FINALIZE
  for _ in range(4):
      K, hop = rng.choice([(401, 160), (401, 160), (201, 80), (801, 320)])
      F = rng.choice([1, 3, 40, 40, 64, 80, 130])
      L = {401: 1600, 201: 1600, 801: 960}[K]
      T = rng.choice([1, hop - 1, L - 1, L, L + 1, 3 * L, 10 * L, 10 * L + 7, rng.randrange(2, 12 * L)])
      B = rng.choice([1, 2, 5, 24, 255, 256, 257, 300, 512])
      if B * T * F > 3.0e9 / 8:
          B = min(B, 24)
      pcen = rng.random() < 0.75
      geo = lo.LeafGeometry(F, 0, K, hop, *lo.same_padding(K))
      params = lo.default_params(geo, pcen, kernel=torch.stack(
          [torch.rand(F, generator=gen) * math.pi, 2.0 + torch.rand(F, generator=gen) * K / 4], dim=1))
      params = {k: v * (1 + 0.1 * (2 * torch.rand(v.shape, generator=gen) - 1)) for k, v in params.items()}
      x = (2 * torch.rand(B, 1, T, generator=gen) - 1).to(DEV)
      m = make_leaf(F, K, hop, pcen, params, DEV)
      print(f"F={F} K={K} hop={hop} T={T} B={B} pcen={pcen}", flush=True)
      with torch.no_grad():
          for name, algo in (("wg", _native.ALGO_FFT_WG), ("stream", stream)):
              m._algo = algo
              t0 = time.time(); y = m(x); torch.cuda.synchronize(); print("   ", name, f"{time.time()-t0:.3f}s", flush=True)
          picks = sorted({0, B - 1, B // 2, rng.randrange(B)})
          m._algo = _native.ALGO_FFT_WG
          t0 = time.time(); y = m(x[picks].contiguous()); torch.cuda.synchronize(); print("    sub", f"{time.time()-t0:.3f}s", flush=True)
          m._algo = _native.ALGO_FFT
          t0 = time.time(); y = m(x[picks].contiguous()); torch.cuda.synchronize(); print("    fft", f"{time.time()-t0:.3f}s", flush=True)
      if T * F * K * len(picks) < 2e8:
          t0 = time.time(); lo.leaf_forward(x[picks].cpu(), params, geo, pcen, torch.float32); print("    oracle", f"{time.time()-t0:.3f}s", flush=True)

import random, time, torch, sys
sys.path.insert(0,'.')
from protocols.distributed_keygen_amd import Engine, limbs as L
from sympy import jacobi_symbol
eng = Engine()
for bits, groups, gs in ((131, 64, 160), (1028, 64, 160), (2053, 64, 160), (4100, 16, 160)):
    rng = random.Random(bits)
    mods = [rng.getrandbits(bits) | (1 << (bits-1)) | 1 for _ in range(groups)]
    vals = [rng.randrange(m) for m in mods for _ in range(gs)]
    limbs = L.limbs_for_bits(bits)
    t = eng.to_device(L.pack(vals, limbs))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = eng.jacobi_t(t, mods, gs); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    got = out.cpu().numpy()
    ok = all(int(got[k]) == jacobi_symbol(vals[k], mods[k // gs]) for k in range(0, len(vals), 97))
    print(bits, groups*gs, f"{dt*1e3:.2f} ms", f"{groups*gs/dt:.0f} /s", ok)

"""Tensor-level test double of the engine on CPU tensors (oracle arithmetic) for the gloo tests."""

from __future__ import annotations

import numpy as np
import torch

from oracle import oracle
from protocols.distributed_keygen_amd import limbs as L


def _ints(t):
    return L.unpack(t.detach().cpu().numpy().view(np.uint32))


def _rows(vals, limbs):
    return torch.from_numpy(L.pack(vals, limbs).view(np.int32))


class FakeTensorEngine:
    def powmod_shared_t(self, bases_t, mod, exp, out_t=None):
        return _rows([oracle.pow_mod(b, exp, mod) for b in _ints(bases_t)], bases_t.shape[1])

    def powmod_nsquare_t(self, bases_t, n, exp, out_t=None):
        return _rows([oracle.pow_mod(b, exp, n * n) for b in _ints(bases_t)], bases_t.shape[1])

    def powmod_multi_t(self, bases_t, mods, exps, group_size, out_t=None):
        vals = _ints(bases_t)
        out = [oracle.pow_mod(b, exps[k // group_size], mods[k // group_size]) for k, b in enumerate(vals)]
        return _rows(out, bases_t.shape[1])

    def sieve_t(self, cands_t, primes, out_t=None):
        return torch.tensor([int(oracle.small_prime_divisors_test(primes, c)) for c in _ints(cands_t)], dtype=torch.uint8)

    def biprime_v_t(self, g_t, mods, exps, group_size, keep):
        limbs = g_t.shape[1]
        vals = _ints(g_t)
        rows, counts = [], []
        for c, (m, e) in enumerate(zip(mods, exps)):
            gs = vals[c * group_size : (c + 1) * group_size]
            kept = [g for g in gs if oracle.jacobi_symbol(g, m) == 1][:keep]
            counts.append(len(kept))
            rows += [oracle.pow_mod(g, e, m) for g in kept] + [oracle.pow_mod(0, e, m)] * (keep - len(kept))
        return _rows(rows, limbs), torch.tensor(counts, dtype=torch.int32)

    def combine_t(self, partials_t, n, theta_inv, out_t=None, status_t=None, packed=False):
        npart, batch, _ = partials_t.shape
        cols = [_ints(partials_t[i]) for i in range(npart)]
        msgs, status = [], []
        for e in range(batch):
            try:
                msgs.append(oracle.decrypt_combine({i + 1: cols[i][e] for i in range(npart)}, n, npart - 1, theta_inv))
                status.append(0)
            except ValueError:
                msgs.append(0)
                status.append(1)
        if packed:
            return torch.cat([_rows(msgs, L.limbs_for(n)), torch.tensor(status, dtype=torch.int32).reshape(-1, 1)], dim=1)
        return _rows(msgs, L.limbs_for(n)), torch.tensor(status, dtype=torch.uint8)

    def biprime_verdict_t(self, v_t, mods, pass_t=None):
        npar, groups, nslots, limbs = v_t.shape
        vals = [_ints(v_t[i].reshape(-1, limbs)) for i in range(npar)]
        out = torch.zeros((groups, nslots), dtype=torch.uint8)
        for g in range(groups):
            for k in range(nslots):
                prod = 1
                for i in range(1, npar):
                    prod *= vals[i][g * nslots + k]
                v1 = vals[0][g * nslots + k]
                out[g, k] = int(v1 % mods[g] == prod % mods[g] or v1 % mods[g] == (-prod) % mods[g])
        return out

"""Worker of tests/test_gpu_distributed.py: two ranks sharing GPU 0 (the test box has one GPU),
process group over gloo. Everything bench.py does for N > 1 except RCCL itself: the key is built
on rank 0, exported as a device blob, broadcast, imported on the other rank; every rank bootstraps
its own index range and checks it against the oracle."""
import os
import sys

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

import concrete_amd.parameters as P
from concrete_amd.backend import Context, DeviceServerKey
from concrete_amd.distributed import replicate_server_key, shard_bounds
from conftest import get_keys
from oracle import cpu_oracle as O

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
p = P.SHORTINT_BY_NAME[sys.argv[2]]
ks = get_keys(p, 64, n_override=8)        # same seed on every rank: same keys for the oracle check
ctx = Context(0)
dk = None
if rank == 0:
    dk = DeviceServerKey.from_standard_keys(ctx, ks.device_params(), ks.bsk, ks.ksk, torus_bits=64)
dk = replicate_server_key(ctx, ks.device_params(), dk)
lut = ks.o.generate_accumulator(lambda x: (x + 1) % 4)
dk.upload_lut(lut)
delta = O.shortint_delta(p)
rng = O.Rng(123)                           # identical inputs on every rank, each takes its shard
total = 13
cts = np.stack([ks.o.lwe_encrypt(rng, ks.glwe, (m % 16) * delta, ks.lwe_std) for m in range(total)])
lo, hi = shard_bounds(total, world, rank)
mine = torch.from_numpy(cts[lo:hi].view(np.int64)).cuda()
out = torch.empty_like(mine)
dk.ks_pbs_device(mine, None, out, hi - lo, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
got = out.cpu().numpy().view(np.uint64)
want = ks.o.ks_pbs_batch(ks.ksk, ks.fbsk, lut, None, cts[lo:hi], threads=2)
if dk.variant == 2:   # wave kernel: decrypt-level parity
    dec = lambda a: [O.shortint_decode(p, ks.o.lwe_phase(ks.glwe, c)) for c in a]
    assert dec(got) == dec(want), (dec(got), dec(want))
else:
    assert np.array_equal(got, want)
# gather every shard on rank 0 to show the pieces tile the batch
pieces = [None] * world
dist.all_gather_object(pieces, (lo, hi, got.tobytes()))
if rank == 0:
    assert [(a, b) for a, b, _ in pieces] == [shard_bounds(total, world, r) for r in range(world)]
print(f"rank {rank} ok variant {dk.variant} shard [{lo},{hi})", flush=True)
dist.destroy_process_group()

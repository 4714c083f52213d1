#!/usr/bin/env python3
"""Carries an instruction-count model (profiles/r0N_instr_model.json) over to a library whose modelled kernels are the
SAME MACHINE CODE as the library the model was fitted to — e.g. after edits of comments, macro names or other kernels.

  adopt_instr_model.py <old model.json> <checkout the model was fitted on> <new model.json>

Checks, in this order: (1) the old model's recorded source digest equals the digest of the kernel sources in <checkout>
(so <checkout> really is the tree the model was fitted to); (2) <checkout>'s built library and the in-tree library have
the same kernel_code_digest (tools/calibrate_instr.py: sha256 over the function bytes of every instance of the three
modexp kernel templates).  Only then writes the new model = old constants + "kernel_code_sha256".  Anything else needs
a fresh fit under rocprofv3 --pmc SQ_INSTS_VALU (tools/profile_round.sh)."""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tools import calibrate_instr as C  # noqa: E402

old_path, checkout, new_path = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3])
model = json.loads(old_path.read_text())
h = hashlib.sha256()
for name in model["kernel_sources"]:
    h.update((checkout / "protocols" / "distributed_keygen_amd" / "csrc" / name).read_bytes())
assert h.hexdigest() == model["kernel_sources_sha256"], "the checkout is not the tree this model was fitted to"
then = C.kernel_code_digest(checkout / "protocols" / "distributed_keygen_amd" / "libmxpaillier.so")
now = C.kernel_code_digest()
assert then == now, f"the modelled kernels' machine code differs ({then[:16]} vs {now[:16]}): fit a new model on the GPU"
model["kernel_code_sha256"] = now
model["kernel_code_of"] = list(C.MODELLED_KERNELS)
model["adopted_from"] = (f"{old_path.name}: constants fitted to SQ_INSTS_VALU on the library of the tree whose kernel sources hash to "
                         f"{model['kernel_sources_sha256'][:16]}; tools/adopt_instr_model.py found the machine code of all modelled kernels "
                         "byte-identical in the current library")
model["kernel_sources_sha256"] = C.kernel_sources_digest()
new_path.write_text(json.dumps(model, indent=0) + "\n")
print(f"{new_path}: kernel_code_sha256 {now}")

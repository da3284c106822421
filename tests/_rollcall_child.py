"""Child process of tests/test_gibbs_gpu.py::test_resident_chain_falls_back_when_its_workgroups_are_not_resident_together: run with a CU mask in the
environment (fewer compute units than the device reports), so that bt_gibbs_noise_chain_begin's residency check — made against the whole GPU — passes for
a launch whose workgroups cannot all be resident.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402
from bayestyper_amd import lib, synth  # noqa: E402

n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000
S = 1
orc = _oracle.load_oracle()
ctx = lib.Ctx(0)
flat = synth.make_batch("A", n_groups, S, seed=5, templates=8)
flat["group_index"] = np.arange(flat["num_groups"], dtype=np.uint32)
lut_g, lut_n = _oracle.build_luts(orc, S)
n_it, first_collect = 5, 2
tables = [_oracle.build_luts(orc, S, noise_rate=0.02 + 0.03 * i)[1] for i in range(n_it)]
kw = dict(seed=77, chains=1, burn=first_collect, iters=n_it - first_collect, noise_seeding=1)
ga, gb = lib.Gibbs(ctx, flat, lut_g, lut_n, **kw), lib.Gibbs(ctx, flat, lut_g, lut_n, **kw)
ga.set_noise_lut(lut_n)
ga.init_chain(0)
want = [ga.noise_iteration(tables[it] if it else None, it >= first_collect) for it in range(n_it)]
gb.set_noise_lut(lut_n)
gb.init_chain(0)
resident = bool(gb.noise_chain_begin(n_it, first_collect))
same = True
if resident:
    for it in range(n_it):
        hb = gb.noise_chain_step(tables[it] if it else None)
        same = same and bool(np.array_equal(want[it], hb)) and int(hb.sum()) > 0
    gb.noise_chain_end()
else:
    for it in range(n_it):
        hb = gb.noise_iteration(tables[it] if it else None, it >= first_collect)
        same = same and bool(np.array_equal(want[it], hb))
ctx.sync()
ra, rb = ga.results(), gb.results()
res_same = all(np.array_equal(ra[k], rb[k]) for k in ra)
# a later chain on this device does not ask for a resident launch again once a roll call has failed
gb.reset_groups()
gb.init_chain(0)
again = bool(gb.noise_chain_begin(n_it, first_collect))
if again:
    gb.noise_chain_step(None)
    gb.noise_chain_end()
ga.close(), gb.close()
print(json.dumps({"resident": resident, "histograms_equal": same, "results_equal": bool(res_same), "resident_again": again, "tiles": (n_groups + 63) // 64}))

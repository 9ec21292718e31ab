"""DistributedDataParallel(VISingerTrainer) with world > 1 on the product's own training path -- two FRESH child ranks
on the one GPU of the box (gloo rendezvous), both optimizer passes with the requires_grad toggle of the reference
(utils/commons/trainer.py:312-375); the all-reduced gradients equal a single-process step on the global batch."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddp_trainer_two_ranks_match_global_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    procs = []
    for r in range(2):      # fresh children (never a re-exec of this GPU-initialised process)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "ddp_two_ranks.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(line) == 1, outs[0]
    rep = json.loads(line[0])
    assert rep["world"] == 2 and rep["weights_equal_across_ranks"] and rep["native_library"] == "libvisinger_hip.so"
    gen, disc = rep["passes"]
    for ps in (gen, disc):
        assert ps["gradients_compared"] > 50 and ps["presence_mismatches"] == 0, ps
        # mean-reduced losses over equal-length items: the average of the two shards' gradients is the global batch's gradient; what is left
        # is fp32 summation order (a shard of 2 items dispatches other tile shapes than the batch of 4)
        assert ps["worst_rel_err"] <= 2e-3, json.dumps(ps)      # (measured: see profiles/r03_ddp_two_ranks.json)
        assert ps["other_network_has_grads"] is False, ps          # the frozen network of each pass collects nothing

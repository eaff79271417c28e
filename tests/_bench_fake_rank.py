"""One rank of bench.py with the engine replaced by the CPU stand-in of test_bench_flow_cpu.py (started by
bench.launch_ranks in the self-launch tests; not a test module itself)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import hgmm_amd  # noqa: E402
import bench  # noqa: E402
from test_bench_flow_cpu import FakeContext  # noqa: E402


class Ctx(FakeContext):
    def comm_init(self, world, rank, uid):
        broken = sys.argv[-1]
        if broken in ("RCCL_BROKEN", "RCCL_AND_IPC_BROKEN") or (broken == "RCCL_BROKEN_ON_RANK_1" and rank == 1):
            raise hgmm_amd.HgmmError("ncclCommInitRank failed (test)")
        if broken == "RCCL_HANGS_ON_RANK_0":
            if rank == 1:
                raise hgmm_amd.HgmmError("ncclCommInitRank failed (test)")
            import time
            time.sleep(3600)                                  # the healthy rank waits for its peer inside the collective call
        super().comm_init(world, rank, uid)

    def comm_init_ipc(self, world, rank, name):
        if sys.argv[-1] == "RCCL_AND_IPC_BROKEN":
            raise hgmm_amd.HgmmError("hipIpcOpenMemHandle failed (test)")
        super().comm_init_ipc(world, rank, name)


if "FAIL_RANK_1" in sys.argv[-1] and os.environ.get("RANK") == "1":
    sys.exit(3)
hgmm_amd.Context = Ctx
bench.N_POINTS = 2000
bench.synth_frame = lambda seed, n=None: np.random.RandomState(seed).rand(2000, 3).astype(np.float32)
if "pairs" in sys.argv:                                        # --mode pairs: the registration itself needs the engine
    class _Tf:
        rot, t = np.eye(3), np.zeros(3)

        def transform(self, x):
            return x

    class _Res:
        transformation = _Tf()

    _cloud = np.random.RandomState(3).rand(50, 3)
    bench.scan_pairs = lambda rank, count=16: (_cloud, [(_cloud + 0.01, _cloud + 1e-4)] * 2)
    bench.register_pair = lambda ctx, source, target: (_Res(), 7)
    bench.register_batch = lambda ctx, source, targets: ([_Res() for _ in targets], [7] * len(targets))
    bench.pairs_cpu_baseline = lambda: {"value": 0.1, "unit": "pairs/s on the sample", "cores": 1, "kind": "port", "sample": "fake"}
bench.main()

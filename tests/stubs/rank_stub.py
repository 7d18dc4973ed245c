"""Stand-in rank program for tests/test_bench_launcher.py: joins the gloo group the launcher's environment
describes, all-reduces a counter and lets rank 0 print one JSON line (what bench.py's ranks do, minus the GPU)."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(7)
dist.init_process_group("gloo")
c = torch.tensor([rank + 1, 10], dtype=torch.int64)
dist.all_reduce(c)
if rank == 0:
    print("noise before the line")
    print(json.dumps({"n_gpus": world, "sum": int(c[0]), "tens": int(c[1]), "local_rank": os.environ["LOCAL_RANK"],
                      "master": os.environ["MASTER_ADDR"]}), flush=True)
dist.barrier()
dist.destroy_process_group()

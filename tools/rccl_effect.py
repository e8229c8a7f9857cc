#!/usr/bin/env python3
"""Does an initialised RCCL communicator slow the single-GPU legs?  usage: python tools/rccl_effect.py [0|1] (1 = init a
one-rank nccl group with device_id first, like bench.py)."""
import os
import socket
import sys
import time

import torch

if os.environ.get('LATE_QUEUES'):   # set AFTER `import torch`, before the first HIP call
    os.environ['GPU_MAX_HW_QUEUES'] = os.environ['LATE_QUEUES']

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynavsr_amd  # noqa: E402
dynavsr_amd.configure_runtime()   # hardware queues for the side streams, before the first HIP call
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if len(sys.argv) > 1 and sys.argv[1] == "1":
    import torch.distributed as tdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    tdist.barrier()
r = bench.inner_step_rate(dev)
print("rccl=%s inner step %.2f ms" % (sys.argv[1] if len(sys.argv) > 1 else "0", r["ms_per_step"]))

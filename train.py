#!/usr/bin/env python3
"""``python train.py --dataset-dir ... --candidate-k 64`` - the reference's training CLI on MI355X kernels.
Multi-GPU: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py ...``"""
from gan_amd.train import main

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""``python AIGgan.py --dataset-dir ... --gen-ckpt ... --out ... --candidate-k 64`` - the reference's sampling CLI."""
from gan_amd.sample import main

if __name__ == "__main__":
    main()

"""Import path of the reference (utilities/pt2bench.py): ``pt_to_bench`` and its CLI."""
import argparse

from gan_amd.export import pt_to_bench  # noqa: F401

if __name__ == "__main__":
    p = argparse.ArgumentParser(description="Convert .pt -> .bench (AIG, inverters as NOT gates)")
    p.add_argument("--input", "-i", required=True)
    p.add_argument("--output", "-o", required=True)
    a = p.parse_args()
    pt_to_bench(a.input, a.output)

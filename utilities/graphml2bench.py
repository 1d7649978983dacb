"""Import path of the reference (utilities/graphml2bench.py): ``convert_graphml_to_bench`` and its CLI."""
import argparse
import os

from gan_amd.export import convert_graphml_to_bench  # noqa: F401


def main(argv=None):
    p = argparse.ArgumentParser(description="Convert GraphML (AIG) to .bench")
    p.add_argument("--graphml", required=True)
    p.add_argument("--out", required=False)
    p.add_argument("--no-repair", action="store_true")
    p.add_argument("--no-diversify", action="store_true")
    a = p.parse_args(argv)
    if not os.path.exists(a.graphml):
        raise FileNotFoundError(a.graphml)
    out = a.out
    if out is None:
        out = os.path.splitext(a.graphml)[0] + ".bench"
    elif os.path.isdir(out):
        out = os.path.join(out, os.path.splitext(os.path.basename(a.graphml))[0] + ".bench")
    convert_graphml_to_bench(a.graphml, out, repair_dag=not a.no_repair, diversify_pad=not a.no_diversify)
    print(f"[OK] Wrote {out}")


if __name__ == "__main__":
    main()

"""Import path of the reference (utilities/scripts_check.py): quick GraphML / .bench checks and its CLI."""
import argparse
import os

from gan_amd.report import check_bench_lines, check_graphml_lines


def check_aig_graphml(graphml_path: str) -> None:
    print("\n".join(check_graphml_lines(graphml_path)))


def check_bench_file(bench_path: str) -> None:
    print("\n".join(check_bench_lines(bench_path)))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Quick checker for GraphML/.bench AIGs")
    ap.add_argument("--graphml", required=True, help="GraphML file to check")
    ap.add_argument("--bench", required=True, help=".bench file to check")
    args = ap.parse_args()
    for path in (args.graphml, args.bench):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
    check_aig_graphml(args.graphml)
    check_bench_file(args.bench)

"""Import path of the reference (utilities/pt2graphml.py): ``pt_to_graphml`` and its CLI."""
import argparse

from gan_amd.export import pt_to_graphml  # noqa: F401


def main():
    p = argparse.ArgumentParser(description="Convert .pt (AIG) to GraphML")
    p.add_argument("-i", "--input", required=True)
    p.add_argument("-o", "--output", required=True)
    a = p.parse_args()
    pt_to_graphml(a.input, a.output)


if __name__ == "__main__":
    main()

"""Entry point of ``python -m vr180_convert_amd``: the command line of cli.py under the reference's program name."""
import sys

from . import cli

if __name__ == "__main__":
    cli.main(sys.argv[1:])

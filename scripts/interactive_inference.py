#!/usr/bin/env python3
"""`python scripts/interactive_inference.py --config_path configs/longlive_interactive_inference.yaml` -- the reference's
interactive_inference.py on the MI355X path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longlive_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main(["interactive"] + sys.argv[1:]))

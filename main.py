#!/usr/bin/env python3
"""Entry point with the reference's command line (main.py:94-123): ``python main.py --foldnum=0 --epoch=1``.
The implementation lives in session-based-news-recommendation_amd/host/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tcar_amd  # noqa: E402,F401
from tcar_amd.host.cli import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])

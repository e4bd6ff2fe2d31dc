"""CLI of magi_v2_amd.isa_check (the guard against vector instructions in front of a join block's EXEC restore):
    python tools/check_exec_prologue.py file.s [...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magi_v2_amd.isa_check import main

if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

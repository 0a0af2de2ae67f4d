"""CPU oracle for the comms-rs DSP hot path -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front end of oracle.cpp (see that file's header).  Importable
only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
the product package (comms_rs_amd/) never imports this.
"""
from .oracle import *  # noqa: F401,F403

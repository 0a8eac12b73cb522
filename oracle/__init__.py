"""CPU oracle loaders — TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; raytracing_amd/ never does."""
import os

from raytracing_amd.sbapi import SbLib

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref.so")


def load_oracle():
    """Plain-C restatement (oracle/rt_oracle.c)."""
    return SbLib(ORACLE_SO)


def have_ref():
    return os.path.exists(REF_SO)


def load_ref():
    """The genuine reference compiled from its own sources (oracle/_ref, built by oracle/Makefile)."""
    return SbLib(REF_SO)

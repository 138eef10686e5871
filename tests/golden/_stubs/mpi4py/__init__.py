"""Throw-away single-rank stand-in so the reference's symbolic layer imports in the
build container (no MPI here).  Used ONLY by tests/golden/make_golden.py."""

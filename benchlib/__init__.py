"""Helpers of bench.py (the driver's contract - one JSON line - lives in bench.py itself)."""

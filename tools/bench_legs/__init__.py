"""Pieces of bench.py that are not the timed region of the headline metric: the compact driver-facing line and the legs
of the other BASELINE configurations / SURVEY section 8 rows.  bench.py (repo root) stays the driver-facing script."""

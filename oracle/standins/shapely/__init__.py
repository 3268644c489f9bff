"""Import-only stand-in for the absent ``shapely`` (TEST INFRASTRUCTURE ONLY).

Reference ``inference.py:10`` imports ``shapely.geometry.Polygon`` at module
scope; the oracle only needs ``inference.find_N_peaks`` (``inference.py:21-29``),
so the geometry classes here just have to be importable.
"""

"""Test-only stand-in for timm==0.4.12 (NOT installed in this image).

Only what /root/reference/models/*.py and models_act.py import.
Used solely by tests/golden/gen_golden.py, in the build container, to import the
reference as the oracle-of-the-oracle.  Never shipped with, nor imported by, the
product package.
"""

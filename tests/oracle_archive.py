"""Imports oracle/reseq_archive.py (the CPU checker of the profile-archive reader; test infrastructure only) as `ra`."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "reseq_archive.py")
_spec = importlib.util.spec_from_file_location("oracle_reseq_archive", _path)
ra = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ra)

import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "rmr_tests_conftest", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "conftest.py"))
_m = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_m)
pytest_configure, kat, oracle = _m.pytest_configure, _m.kat, _m.oracle

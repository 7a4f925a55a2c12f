"""Unit helpers used on the way into the path (same results as simfire/utils/units.py)."""
import re
from datetime import timedelta

_UNITS = {"s": "seconds", "m": "minutes", "h": "hours", "d": "days", "w": "weeks"}


def mph_to_ftpm(mph):
    """mph -> ft/min (simfire/utils/units.py:34-45)."""
    return mph * 88


def ftpm_to_mph(ftpm):
    return ftpm / 88


def meters_to_feet(meters):
    """simfire/utils/units.py:88-100"""
    return meters * 3.28084


def str_to_minutes(string: str) -> int:
    """'1d 2h 3m' / '24h' / '90' -> minutes (simfire/utils/units.py:62-85): every
    ``<number><unit>`` group is summed, a bare number counts as minutes."""
    kw = {}
    for m in re.finditer(r"(?P<val>\d+(\.\d+)?)(?P<unit>[smhdw]?)", str(string), flags=re.I):
        kw[_UNITS.get(m.group("unit").lower(), "minutes")] = float(m.group("val"))
    return int(round(timedelta(**kw).total_seconds() / 60))

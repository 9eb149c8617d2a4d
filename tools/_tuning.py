"""Tool-side helper: kernel-selection fields of sanerf_hq_amd.raymarching.Tuning from a string "field=value,field=value"
(command-line flag --tuning of the A/B tools, or the SN_TUNING variable READ BY THE TOOLS -- the library itself reads no environment)."""
import os


def parse(text):
    from sanerf_hq_amd import raymarching as rm
    kw = {}
    for item in (text or "").split(","):
        item = item.strip()
        if item:
            k, v = item.split("=")
            kw[k.strip()] = int(v)
    return rm.Tuning(**kw)


def apply_default(text=None):
    """Set the process default (raymarching.tuning) from `text` or $SN_TUNING; returns the string applied."""
    from sanerf_hq_amd import raymarching as rm
    text = os.environ.get("SN_TUNING", "") if text is None else text
    t = parse(text)
    for f in rm.Tuning.FIELDS:
        setattr(rm.tuning, f, getattr(t, f))
    return text

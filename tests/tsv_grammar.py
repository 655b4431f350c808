"""TSV grammar of sigtk's per-record subtools (reference: src/cfunc.c, src/jnn.c:309-350).

TEST INFRASTRUCTURE.  Python mirror of the printers so that results coming back through the
C-ABI (or from the test oracle) can be compared byte-for-byte with the reference CLI's
stdout.  The product CLI (sigtk_amd/host) prints from C; nothing of the product imports this.

``%f`` of a float32 is printed by C after promotion to double; Python's ``'%f' % float(x)``
formats the identical double with correct rounding, so the bytes agree with glibc.
"""
from __future__ import annotations

import numpy as np

HDR_EVENT = "read_id\tevent_idx\traw_start\traw_end\tevent_mean\tevent_std\n"          # cfunc.c:67
HDR_EVENT_COMPACT = "read_id\tlen_raw_signal\traw_start\traw_end\tnum_event\tevents\n"  # cfunc.c:65
HDR_PA = "read_id\tlen_raw_signal\tpa\n"                                                 # cfunc.c:105
HDR_JNN = "read_id\tlen_raw_signal\tnum_seg\tseg\n"                                      # cfunc.c:119
HDR_STAT = ("read_id\tlen_raw_signal\traw_mean\tpa_mean\traw_std\tpa_std\traw_median\tpa_median\n")  # cfunc.c:123
HDR_PREFIX = "read_id\tlen_raw_signal\tadapt_start\tadapt_end\tpolya_start\tpolya_end"  # cfunc.c:162
HDR_PREFIX_STAT = "\tadapt_mean\tadapt_std\tadapt_median\tpolya_mean\tpolya_std\tpolya_median"  # cfunc.c:164


def _f(x) -> str:
    return "%f" % float(x)


def event_hdr(compact: bool) -> str:
    return HDR_EVENT_COMPACT if compact else HDR_EVENT


def prefix_hdr(p_stat: bool) -> str:
    return HDR_PREFIX + (HDR_PREFIX_STAT if p_stat else "") + "\n"


def event_rows(read_id: str, n: int, start, length, mean, stdv, compact: bool) -> str:
    """print_events (cfunc.c:16-61).  length may be float32 (reference) or integer."""
    start = np.asarray(start).astype(np.int64)
    ilen = np.asarray(length).astype(np.int64)  # (int)length
    nev = start.size
    if compact:
        out = ["%s\t%d\t" % (read_id, n)]
        if nev:
            out.append("%d\t%d\t" % (start[0], start[-1] + ilen[-1]))
            out.append("%d\t" % nev)
            parts = []
            for j in range(nev):
                if ilen[j]:
                    parts.append(("%d," if j < nev - 1 else "%d") % ilen[j])
            out.append("".join(parts))
        else:
            out.append(".\t.\t.\t.")
        out.append("\n")
        return "".join(out)
    rows = ["%s\t%d\t%d\t%d\t%s\t%s\n" % (read_id, j, start[j], start[j] + ilen[j], _f(mean[j]), _f(stdv[j]))
            for j in range(nev)]
    rows.append("\n")  # cfunc.c:58: an empty line after every read
    return "".join(rows)


def pa_row(read_id: str, pa) -> str:
    """pa_func (cfunc.c:85-102)."""
    pa = np.asarray(pa, dtype=np.float32)
    return "%s\t%d\t%s\n" % (read_id, pa.size, ",".join(_f(v) for v in pa))


def stat_row(read_id: str, n: int, raw_mean, pa_mean, raw_std, pa_std, raw_median, pa_median) -> str:
    """stat_func (cfunc.c:126-159); note the trailing tab."""
    return "%s\t%d\t%s\t%s\t%s\t%s\t%d\t%s\t\n" % (read_id, n, _f(raw_mean), _f(pa_mean), _f(raw_std),
                                                   _f(pa_std), int(raw_median), _f(pa_median))


def jnn_row(read_id: str, n: int, x, y, compact: bool) -> str:
    """jnn_func + jnn_print (cfunc.c:108-117, jnn.c:309-350).

    The reference prints nothing after the length column when the read is empty
    (jnn_raw returns NULL); otherwise ``num_seg\\t`` then the segment list or ``.``."""
    x = np.asarray(x).astype(np.int64)
    y = np.asarray(y).astype(np.int64)
    out = ["%s\t%d\t" % (read_id, n)]
    if n > 0:
        out.append("%d\t" % x.size)
        if compact:
            ci = 0
            for i in range(x.size):
                mi = int(x[i]) - ci
                ci += mi
                if mi:
                    out.append("%dH" % mi)
                mi = int(y[i]) - ci
                ci += mi
                if mi:
                    out.append("%d," % mi)
        else:
            for i in range(x.size):
                out.append("%d,%d;" % (x[i], y[i]))
        if x.size == 0:
            out.append(".")
    out.append("\n")
    return "".join(out)


def prefix_row(read_id: str, n: int, adapt, polya, p_stat: bool, adapt_stats=None, polya_stats=None) -> str:
    """prefix_func (cfunc.c:169-234).

    adapt = (x, y) as find_adaptor returns it; polya = (x, y) RELATIVE to adapt y as
    find_polya returns it (or (-1,-1)).  Quirks kept: with --print-stat a double tab
    precedes the polyA statistics and the stats end with a tab (cfunc.c:204,210)."""
    ax, ay = int(adapt[0]), int(adapt[1])
    out = ["%s\t%d\t" % (read_id, n)]
    if ay > 0:
        out.append("%d\t%d\t" % (ax, ay))
        px, py = int(polya[0]), int(polya[1])
        if py > 0:
            out.append("%d\t%d" % (px + ay, py + ay))
        else:
            out.append(".\t.")
        if p_stat:
            out.append("\t%s\t%s\t%s\t" % tuple(_f(v) for v in adapt_stats))
            if py > 0:
                out.append("\t%s\t%s\t%s\t" % tuple(_f(v) for v in polya_stats))
            else:
                out.append("\t.\t.\t.")
    else:
        out.append(".\t.\t.\t.")
    out.append("\n")
    return "".join(out)


# ---- ent (src/ent.c:105, :108-163): id, then "%f" of the three entropies (doubles)
HDR_ENT = "read_id\traw_ent\tdelta_ent\tbyte_ent\n"


def ent_row(read_id: str, raw_ent: float, delta_ent: float, byte_ent: float) -> str:
    return "%s\t%f\t%f\t%f\n" % (read_id, raw_ent, delta_ent, byte_ent)

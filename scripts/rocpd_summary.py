"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this turns one into the CSV summaries kept under profiles/.
   python scripts/rocpd_summary.py stats  <results.db> <out.csv>      per-kernel calls / total / average / min / max (ns)
   python scripts/rocpd_summary.py pmc    <results.db> <out.csv>      per-kernel mean of every collected counter per launch
   python scripts/rocpd_summary.py hbm    <fetch.db> <write.db> <out.csv>   HBM bytes per launch from the FETCH_SIZE and the WRITE_SIZE pass
   python scripts/rocpd_summary.py timeline <results.db> <out.csv> [first kernel name part]   the LAST repetition of a launch sequence
                                          (from the last launch whose name holds that part, default k_fk): start / end offsets in us
"""
import csv
import sqlite3
import sys


def short(name):
    return name if len(name) < 120 else name[:60] + " ... " + name[-40:]


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r[0]), r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])


def pmc(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    names = sorted(set(r[1] for r in rows))
    per = {}
    for k, n, cnt, v, dur in rows:
        per.setdefault(k, {"launches": cnt, "avg_ns": dur})[n] = v
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        # derived (MI355X_MICROARCH.md: SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles summed over the chip's 1024 SIMDs,
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs):  valu_busy = 4*SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs);
        # waves_per_simd = 4*SQ_WAVE_CYCLES / the same denominator; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES
        w.writerow(["kernel", "launches", "avg_duration_ns"] + names + ["valu_busy", "waves_per_simd", "wait_frac", "clock_GHz"])
        for k, d in sorted(per.items(), key=lambda kv: -(kv[1]["avg_ns"] or 0) * kv[1]["launches"]):
            extra = ["", "", "", ""]
            if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_WAVE_CYCLES"):
                den = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
                extra = ["%.3f" % (4 * d.get("SQ_ACTIVE_INST_VALU", 0) / den), "%.2f" % (4 * d["SQ_WAVE_CYCLES"] / den),
                         "%.2f" % (d.get("SQ_WAIT_ANY", 0) / d["SQ_WAVE_CYCLES"]), "%.2f" % (d["GRBM_GUI_ACTIVE"] / 8.0 / max(d["avg_ns"], 1))]
            w.writerow([short(k), d["launches"], "%.0f" % (d["avg_ns"] or 0)] + ["%.1f" % d.get(n, float("nan")) for n in names] + extra)


def hbm(db_fetch, db_write, out):
    """MI355X_MICROARCH.md (HBM): FETCH_SIZE and WRITE_SIZE in separate passes, counters in KB, gfx950 correction
    bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
    per = {}
    for db, name in ((db_fetch, "FETCH_SIZE"), (db_write, "WRITE_SIZE")):
        c = sqlite3.connect(db)
        for k, cnt, v in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (name,)):
            per.setdefault(k, {})[name] = v
            per[k]["launches"] = cnt
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_bytes_per_launch"])
        for k, d in sorted(per.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
            fe, wr = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
            w.writerow([short(k), d["launches"], "%.1f" % fe, "%.1f" % wr, "%.0f" % ((2 * fe + wr) * 1024)])


def timeline(db, out, first="k_fk"):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    if len(idx) >= 2:       # the last complete repetition
        rows = rows[idx[-2]:idx[-1]]
    t0 = rows[0][1]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "start_us", "end_us", "duration_us", "gap_before_us"])
        last_end = t0
        for n, s_, e_ in rows:
            w.writerow([short(n)[:70], "%.1f" % ((s_ - t0) / 1e3), "%.1f" % ((e_ - t0) / 1e3), "%.1f" % ((e_ - s_) / 1e3), "%.1f" % ((s_ - last_end) / 1e3)])
            last_end = max(last_end, e_)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "hbm": hbm, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])

#!/usr/bin/env python3
"""The kernel table of DESIGN.md section 4 from profiles/rNN_pmc_all.json (tools/pmc_all.sh + tools/assemble_profiles.py):
one row per kernel with its algorithmic bytes, instructions per unit, traffic by counters over algorithmic, the in-kernel
clock, the rate and both roofline fractions.  Usage: python tools/pmc_table.py profiles/r04_pmc_all.json [md|txt]
(profiles/r06_kernel_table.md is this script's output on profiles/r06_pmc_all.json; DESIGN.md section 4 includes it verbatim and
tests/test_profiles_reproduce.py re-runs the script and compares)"""
import json
import sys

NAMES = {"band_rx_real_f32": "`band_kernel<rx_real>`", "band_sync_cplx_f32": "`band_kernel<sync_cplx>`",
         "band_dechirp_down_f32": "`band_kernel<dechirp_down>` (frame pairs)", "compress_f32": "`compress_kernel` (frame pairs)",
         "iq1024_fw_f32": "`iq1024_kernel` firmware windows", "iq1024_bb_f32": "`iq1024_kernel` base band (configs[2])",
         "iq2048_fw_f32": "`iq_kernel` n = 2048 firmware windows", "iq2048_bb_f32": "`iq_kernel` n = 2048 base band",
         "stream_d8_f32": "`stream_kernel<8>` (per input sample)",
         "sinc5": "`sinc5_kernel`, one recorded stream (per 32-bit word)",
         "sinc5_streams": "`sinc5_kernel`, one 2048-word block of each of 131 072 microphones (per word)",
         "rows_rx_real_f32": "`band_kernel<rx_real, ROWS>`, every new FIFO offset of a live stream evaluated (per offset)",
         "rows_sync_cplx_f32": "`band_kernel<sync_cplx, ROWS>`, every offset, both references (per offset)",
         "rows_rx_real_f32_idle": "`band_kernel<rx_real, ROWS>`, idle streams: 3 or 5 of the 8 offsets (per offset INDEX)",
         "rows_sync_cplx_f32_idle": "`band_kernel<sync_cplx, ROWS>`, idle streams: 3 or 5 of the 8, UP only (per offset INDEX)",
         "rows_rx_real_f32_idle_keep": "... rx_real idle, `uc_rx_state_keep_previous` (per offset INDEX)",
         "rows_sync_cplx_f32_idle_keep": "... sync_cplx idle, `uc_rx_state_keep_previous` (per offset INDEX)",
         "spectrum_rx_real_f32": "`band_kernel<rx_real, SPEC>` = `uc_window_spectrum` (statistics + 2 x 313 window bins stored)"}
ORDER = list(NAMES)


def table(d, md=True):
    """the rows of the kernel table (a list of lines) for the entries of a *_pmc_all.json that this table knows"""
    out = []
    if md:
        out.append("| kernel | algorithmic B/unit | VALU / LDS wave-instr. per unit | HBM bytes by counters ÷ algorithmic | in-kernel clock | rate (units/s) | **valu** | **hbm** | waves: issue / wait-to-issue / waitcnt |")
        out.append("|---|---|---|---|---|---|---|---|---|")
    for k in ORDER:
        if k not in d:
            continue
        v = d[k]
        e, cp = v["derived"], v.get("clock_probe") or {}
        alg = v["alg_bytes_per_unit"]
        ghz = cp.get("shader_clock_MHz_median", 0.0) / 1e3
        ms, units = cp.get("ms_last_launch_events"), cp.get("units")
        rate = units / (ms * 1e-3) if ms else None
        hbm = rate * alg / 8e12 if rate else None
        valu = e["SQ_INSTS_VALU_per_unit"] * units * 4.0 / (4.0 * 256 * ghz * 1e9 * ms * 1e-3) if rate and ghz else None
        fmt = lambda x, f: (f % x) if x is not None else "—"
        waves = "%.2f / %.2f / %.2f" % (e["SQ_ACTIVE_INST_ANY_over_WAVE_CYCLES"], e["SQ_WAIT_INST_ANY_over_WAVE_CYCLES"],
                                        e["SQ_WAIT_ANY_over_WAVE_CYCLES"])
        vi, li = e["SQ_INSTS_VALU_per_unit"], e["SQ_INSTS_LDS_per_unit"]
        instr = ("%.0f / %.0f" % (vi, li)) if vi >= 10 else ("%.2f / %.2f" % (vi, li))
        row = [NAMES[k], "%g" % alg, instr, "%.4f" % e["hbm_over_algorithmic"], fmt(ghz or None, "%.2f GHz"), fmt(rate, "%.3g"),
               fmt(valu, "%.2f"), fmt(hbm, "%.2f"), waves]
        if md:
            big = 6 if (valu or 0) >= (hbm or 0) else 7
            if row[big] != "—":
                row[big] = "**%s**" % row[big]
            out.append("| " + " | ".join(row) + " |")
        else:
            out.append("%-46s alg %7s B  instr %-12s hbm/alg %s  clock %-9s rate %-9s valu %-5s hbm %-5s  waves %s" % tuple(row))
    return out


if __name__ == "__main__":
    print("\n".join(table(json.load(open(sys.argv[1])), len(sys.argv) < 3 or sys.argv[2] == "md")))

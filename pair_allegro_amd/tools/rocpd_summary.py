"""Summarise a rocprofv3 rocpd database (kernel trace) as a markdown table: python -m pair_allegro_amd.tools.rocpd_summary <db> [top]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("| kernel | calls | total ms | avg ms | % |")   # top_kernels durations are in microseconds
    print("|---|---|---|---|---|")
    for name, calls, tot, avg, pct in rows[:top]:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 70:
            short = short[:67] + "..."
        print(f"| `{short}` | {calls} | {tot/1e3:.3f} | {avg/1e3:.4f} | {pct:.2f} |")
    k = db.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x "
                   "from kernels where name like '%k_fused%' limit 1").fetchall()
    for r in k:
        print(f"\n`k_fused` dispatch: vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]} B scratch={r[5]} B/lane grid={r[6]} block={r[7]}")


if __name__ == "__main__":
    main()

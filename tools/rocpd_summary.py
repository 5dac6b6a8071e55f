"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into a short per-kernel CSV for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/rNN_name.csv "command line that was profiled"
"""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write(f"# {note}\n# source: view top_kernels of {db_path} (durations in microseconds... see unit note)\n")
        f.write("# columns: kernel,calls,total_ns_div_1000,avg_us,percent\n")
        for name, calls, total, avg, pct in rows[:30]:
            short = name if len(name) <= 140 else name[:137] + "..."
            f.write(f"\"{short}\",{calls},{total:.3f},{avg:.3f},{pct:.3f}\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(*sys.argv[1:4])

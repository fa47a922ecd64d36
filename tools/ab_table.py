"""Tabulate a tools/ab_modes.sh log: one row per shape, one column per (variant, mode)."""
import re
import sys
rows, cols, key = {}, [], None
for l in open(sys.argv[1]):
    m = re.match(r"== variant (\d+) (\S+)", l)
    if m:
        key = (m.group(2), int(m.group(1)))
        cols.append(key)
        continue
    m = re.match(r"(cin.*?):\s+([\d.]+) ms", l)
    if m:
        rows.setdefault(m.group(1), {})[key] = float(m.group(2))
print(" " * 52 + " ".join(f"{c[0][:6]}/v{c[1]}".rjust(10) for c in cols))
for k, d in rows.items():
    print(k.ljust(52) + " ".join(f"{d.get(c, 0):10.3f}" for c in cols))

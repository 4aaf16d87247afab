#!/usr/bin/env python3
"""Extracts the conformance DATA of the reference (item names, ITU-R BS.1387 DI/ODG, and the DI the
reference implementation itself reaches) into tests/golden/conformance_tables.json.

Sources (data only, read in the build container; /root/reference does not travel):
  doc/conformance_{basic,advanced}_table.xml   item | ITU DI | reference's actual DI | difference
  doc/make_conformance_tables.sh:62-77,110-125 item | ITU DI | ITU ODG
"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "conformance_tables.json"

tables = {}
for mode in ("basic", "advanced"):
    xml = (REF / "doc" / f"conformance_{mode}_table.xml").read_text()
    rows = re.findall(r"<entry>([a-z0-9]{7})</entry><entry>([-.0-9]+)</entry><entry>([-.0-9]+)</entry>"
                      r"<entry>([-.0-9]+)</entry>", xml)
    sh = (REF / "doc" / "make_conformance_tables.sh").read_text()
    itu = {m[0]: (m[1], m[2]) for m in
           re.findall(rf"runpeaq --{mode} \"\${{DATADIR}}/([a-z0-9]{{7}})\.wav\"\s+([-.0-9]+)\s+([-.0-9]+)", sh)}
    items = []
    for name, itu_di, actual_di, _diff in rows:
        assert itu[name][0] == itu_di, (name, itu[name], itu_di)
        items.append(dict(item=name, itu_di=itu_di, itu_odg=itu[name][1], reference_di=actual_di))
    assert len(items) == 16, len(items)
    tables[mode] = items
OUT.write_text(json.dumps(tables, indent=1) + "\n")
print(f"wrote {OUT}: {[len(v) for v in tables.values()]}")

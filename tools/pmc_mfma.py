"""MFMA utilisation per kernel (and per launch geometry) from a rocprofv3 --pmc counter_collection.csv that holds
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (optionally SQ_LDS_BANK_CONFLICT, SQ_BUSY_CU_CYCLES):
    MfmaUtil % = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMDs) * 100      (rocprofiler-sdk counter_defs.yaml,
    the gfx950 definition of the derived counter MfmaUtil; SIMDs = 256 CUs x 4)
    python tools/pmc_mfma.py <counter_collection.csv> [min_calls]"""
import collections
import csv
import re
import sys

SIMDS = 256 * 4
XCDS = 8   # GRBM_GUI_ACTIVE is reported summed over the XCDs on gfx950 (calibrated on a GEMM with a known MFMA count)


def main():
    path = sys.argv[1]
    agg = collections.OrderedDict()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = (re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70], r.get('Grid_Size', ''), r.get('Workgroup_Size', ''))
            d = agg.setdefault(k, collections.defaultdict(float))
            d[r['Counter_Name']] += float(r['Counter_Value'])
            d['_rows_' + r['Counter_Name']] += 1
    print('# %s : MfmaUtil = MFMA_BUSY / (GUI_ACTIVE / %d XCDs * %d SIMDs); gui_cyc/call per XCD' % (path.split('/')[-1], XCDS, SIMDS))
    print('%-70s %10s %6s %7s %12s %10s %12s' % ('kernel', 'grid', 'wg', 'calls', 'gui_cyc/call', 'MfmaUtil%', 'lds_confl/call'))
    rows = []
    for (k, g, w), d in agg.items():
        n = d.get('_rows_GRBM_GUI_ACTIVE', 0)
        if not n or d.get('GRBM_GUI_ACTIVE', 0) <= 0:
            continue
        util = 100.0 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (d['GRBM_GUI_ACTIVE'] / XCDS * SIMDS)
        rows.append((d['GRBM_GUI_ACTIVE'], k, g, w, int(n), d['GRBM_GUI_ACTIVE'] / n / XCDS, util, d.get('SQ_LDS_BANK_CONFLICT', 0.0) / n))
    for _, k, g, w, n, cyc, util, conf in sorted(rows, reverse=True):
        if n >= (int(sys.argv[2]) if len(sys.argv) > 2 else 1):
            print('%-70s %10s %6s %7d %12.0f %10.2f %12.0f' % (k, g, w, n, cyc, util, conf))


if __name__ == '__main__':
    main()

"""The StreamStatsTracker restatement (oracle/stats.py) against the vectors the reference itself produced."""
import os

import numpy as np
import pytest

from oracle.stats import StreamStats


@pytest.mark.parametrize("name,beams", [("all", None), ("b32", 32)])
def test_stream_stats_oracle_matches_reference(golden_dir, name, beams):
    g = np.load(os.path.join(golden_dir, "stream_stats.npz"))
    st = StreamStats(use_beams_num=beams)
    for k in range(len(g["scans"])):
        for i in range(10 * k, 10 * (k + 1)):
            st.track_imu(g["imu_a"][i], g["imu_w"][i], float(g["imu_ts"][i]))
        st.track_scan(g["scans"][k], int(g["scan_ts_ns"][k]))
        assert np.array_equal(st.row(), g[f"rows_{name}"][k])  # same numpy operations: bit-exact

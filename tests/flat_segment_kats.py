"""The reference's own flat-segment tests (internal/segment/flat/{segment,quantization,partitioned}_test.go, as data in
tests/golden/reference_kats.json `flat_segment_search`) run against any implementation of three calls:
  search(rows, query, k)                         -> (ids, scores)   fp32 Segment.Search
  search_sq8_rerank(rows, query, k)              -> (ids, scores)   SQ8 Segment.Search, then Segment.Rerank of what it returned
  search_probed(rows, centroids, offsets, query, k, nprobes) -> (ids, scores), rows grouped by partition
The partitioned case needs k-means partitions: `partition(rows, parts)` -> (centroids, offsets, grouped rows)."""
import json
from pathlib import Path

import numpy as np

CASES = json.loads((Path(__file__).resolve().parent / "golden" / "reference_kats.json").read_text())["flat_segment_search"]["cases"]


def run(search, search_sq8_rerank, partition, search_probed):
    seen = 0
    for c in CASES:
        if c["name"] == "TestPartitionedSegment":
            rng = np.random.default_rng(20260403)
            rows = np.array([[10 * (i % 4) + rng.random(), 10 * (i % 4) + rng.random()] for i in range(100)], np.float32)
            cent, off, grouped = partition(rows, c["partitions"])
            assert cent.shape == (c["partitions"], 2) and off.size == c["partitions"] + 1 and off[-1] == 100   # :54-55
            for q in c["queries"]:
                ids, sc = search_probed(grouped, cent, off, np.array(q, np.float32), c["k"], c["nprobes"])
                assert ids.size > 0 and np.all(sc < c["expect_every_score_below"]), (q, sc)
            seen += 1
            continue
        rows = np.array(c["rows"], np.float32)
        q = np.array(c["query"], np.float32)
        if c["quantization"] == "sq8":
            ids, sc = search_sq8_rerank(rows, q, c["k"])
        else:
            ids, sc = search(rows, q, c["k"])
        assert ids.size == c["expect_results"], c["name"]
        best = int(np.argmin(sc))
        assert ids[best] == c["expect_best_row"], (c["name"], ids, sc)
        if "expect_best_score_below" in c:
            assert sc[best] < c["expect_best_score_below"]
        seen += 1
    assert seen == 3

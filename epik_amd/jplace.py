"""jplace v3 reading/writing on the Python side (the C++ writer is epik_amd/host/jplace.cpp).

Field order and document structure follow the reference writer
(epik/src/epik/jplace.cpp:40-158): metadata.invocation, tree, version 3, fields
[edge_num, likelihood, like_weight_ratio, distal_length, pendant_length], then one object
per unique sequence with "p" rows and "nm" [name, 1] pairs."""
from __future__ import annotations

import json

FIELDS = ["edge_num", "likelihood", "like_weight_ratio", "distal_length", "pendant_length"]


def write_jplace(path: str, placed, invocation: str, newick_tree: str) -> None:
    """`placed` is an epik_amd.placer.PlacedCollection (or a list of them, one per batch)."""
    batches = placed if isinstance(placed, (list, tuple)) else [placed]
    placements = []
    for batch in batches:
        for seq in batch.placed_seqs:
            placements.append({
                "p": [[p.branch_id, p.score, p.weight_ratio, p.distal_length, p.pendant_length]
                      for p in seq.placements],
                "nm": [[h, 1] for h in batch.sequence_map[seq.sequence]],
            })
    doc = {"metadata": {"invocation": invocation}, "tree": newick_tree, "version": 3,
           "fields": FIELDS, "placements": placements}
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)


def read_jplace(path: str) -> dict:
    """name -> list of rows (dicts keyed by the file's own `fields`)."""
    with open(path) as fh:
        doc = json.load(fh)
    fields = doc["fields"]
    out = {}
    for obj in doc["placements"]:
        rows = [dict(zip(fields, row)) for row in obj["p"]]
        names = [nm[0] for nm in obj.get("nm", [])] + list(obj.get("n", []))
        for name in names:
            out[name] = rows
    return out

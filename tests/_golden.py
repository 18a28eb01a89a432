"""Helpers to read tests/golden (data captured from the reference; see tests/golden/README.md)."""
import json
import os

import numpy as np

from zotmer_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def load_case(name):
    """(info dict, kmers u64[], counts u64[], raw kmers bytes, raw counts bytes)"""
    info = load_json(name)
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return info, z["kmers"], z["counts"], z["raw_kmers"].tobytes(), z["raw_counts"].tobytes()


def synth_reads(info):
    """Read sequences (list of str) for a case generated from zotmer_amd.synth parameters."""
    return synth.read_strings(**info["synth"])


def synth_fastq(info):
    return synth.fastq_text(**info["synth"])


def hist_dict(counts):
    v, f = np.unique(np.asarray(counts, dtype=np.uint64), return_counts=True)
    return {str(int(a)): int(b) for a, b in zip(v, f)}


def fastq_seqs(text):
    """Sequences of a FASTQ text as file.readFastq sees them (library/file.py:38-52): groups of
    four stripped lines, a trailing partial group is dropped."""
    lines = text.splitlines()
    return [lines[i + 1].strip() for i in range(0, len(lines) - 3, 4)]


def fasta_seqs(text):
    """Sequences of a FASTA text as file.readFasta sees them (library/file.py:19-36)."""
    out, cur, have = [], [], False
    for l in text.splitlines():
        l = l.strip()
        if l[:1] == ">":
            if have:
                out.append("".join(cur))
            have, cur = True, []
        else:
            cur.append(l)
    if have:
        out.append("".join(cur))
    return out


KMERIZE_SYNTH_CASES = ["g2_kmerize_uniformN", "g3_kmerize_genome", "g3_kmerize_genome_k24",
                       "g3_kmerize_genome_k12", "g8_kmerize_k31"]

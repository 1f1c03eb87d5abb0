"""Synthetic BAM / BED writers for tests and examples (pure Python + zlib; htslib/samtools are not available).

Writes spec-conformant BGZF-compressed BAM (SAM spec sections 4.1-4.2) from a records.RecordBatch so that the
`strling` CLI's own BGZF/BAM reader can be exercised end to end.  No index is written: the CLI revisits the
unmapped tail by itself (it does not need `query("*")`).
"""
import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data, level=1):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def sam_header(targets, sort_order="coordinate"):
    return f"@HD\tVN:1.6\tSO:{sort_order}\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in targets)


def write_bam(path, rec, header_text=None, level=1, block=0xFF00):
    targets = rec.targets
    text = (header_text if header_text is not None else sam_header(targets)).encode()
    out = bytearray()
    out += b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(targets))
    for name, length in targets:
        nb = name.encode() + b"\0"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", length)
    isize = rec.isize if rec.isize is not None else np.zeros(rec.n, np.int32)
    for i in range(rec.n):
        qn = rec.qname(i) + b"\0"
        c0, c1 = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
        l_seq = int(rec.l_seq[i])
        so = int(rec.seq_off[i])
        seq = bytes(rec.seq4[so:so + (l_seq + 1) // 2])
        if l_seq & 1 and seq:
            seq = seq[:-1] + bytes([seq[-1] & 0xF0])
        cig = rec.cigar[c0:c1].astype("<u4").tobytes()
        qual = b"\xff" * l_seq
        body = struct.pack("<iiBBHHHiiii", int(rec.tid[i]), int(rec.pos[i]), len(qn), int(rec.mapq[i]), 4680, c1 - c0,
                           int(rec.flag[i]), l_seq, int(rec.mtid[i]), int(rec.mpos[i]), int(isize[i])) + qn + cig + seq + qual
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for o in range(0, len(out), block):
            f.write(_bgzf_block(bytes(out[o:o + block]), level))
        f.write(_EOF)
    return text.decode()


def write_genome_bed(path, genome, targets, unit="AC"):
    with open(path, "w") as f:
        for t in range(genome.n_tid):
            for j in range(int(genome.iv_off[t]), int(genome.iv_off[t + 1])):
                f.write(f"{targets[t][0]}\t{int(genome.iv_start[j])}\t{int(genome.iv_stop[j])}\t{unit}\n")

"""The `strling` CLI: BGZF/BAM reader round trip (CPU) and extract/merge end to end against the oracle (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from strling_amd import api, bamio, build, synth
from strling_amd.records import CIGAR_OPS

CLI = build.CLI


def _run(args, **kw):
    return subprocess.run([CLI] + args, capture_output=True, text=True, **kw)


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    rec, g = synth.synth_wgs(6000, seed=21, contig_len=800_000)
    bam = str(d / "s.bam")
    hdr = bamio.write_bam(bam, rec)
    bed = str(d / "ref.fa.str")
    bamio.write_genome_bed(bed, g, rec.targets)
    return dict(dir=d, rec=rec, g=g, bam=bam, bed=bed, hdr=hdr)


def test_cli_exists_and_help():
    assert os.path.exists(CLI), "strling CLI not built (python -m strling_amd.build)"
    r = _run(["extract"])
    assert r.returncode == 0 and "strling extract" in r.stdout
    r = _run(["pull_region", "x", "y"])
    assert r.returncode == 1 and "not part of this build" in r.stderr


def test_bam_reader_roundtrip(sample):
    """own BGZF/BAM decoder == what the Python writer put in (every field extract.nim reads through hts-nim)"""
    rec = sample["rec"]
    r = _run(["_dump", sample["bam"]])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    nh = len(sample["hdr"].rstrip("\n").split("\n"))
    assert "\n".join(lines[:nh]) + "\n" == sample["hdr"]
    body = [l for l in lines[nh:] if l]
    assert len(body) == rec.n
    for i in list(range(0, rec.n, 97)) + [rec.n - 1]:
        f = body[i].split("\t")
        c0, c1 = int(rec.cigar_off[i]), int(rec.cigar_off[i + 1])
        cig = "".join(f"{int(c) >> 4}{CIGAR_OPS[int(c) & 15]}" for c in rec.cigar[c0:c1]) or "*"
        assert f[0] == rec.qname(i).decode() and int(f[1]) == rec.flag[i] and int(f[2]) == rec.tid[i] and int(f[3]) == rec.pos[i]
        assert int(f[4]) == rec.mapq[i] and f[5] == cig and int(f[6]) == rec.mtid[i] and int(f[7]) == rec.mpos[i]
        assert int(f[8]) == rec.isize[i] and f[9] == rec.sequence(i)


@pytest.mark.parametrize("threads,blocks,batch", [("1", "0", "4096"), ("4", "3", "1000"), ("7", "1", "77"), ("3", "0", "100000")])
def test_parallel_stream_reader_equals_plain_reader(sample, threads, blocks, batch):
    """the multi-threaded whole-file reader (parallel inflate + parallel parse, superchunk carry) yields the same records"""
    a = _run(["_dump", sample["bam"]])
    env = dict(os.environ, STRL_THREADS=threads, STRL_CHUNK_BLOCKS=blocks)
    b = _run(["_dump", sample["bam"], "stream", batch], env=env)
    assert a.returncode == 0 and b.returncode == 0, b.stderr
    assert a.stdout == b.stdout and a.stdout.count("\n") > sample["rec"].n


@pytest.mark.parametrize("block,threads,chunk", [(61, "5", "7"), (100, "3", "1"), (333, "8", "2"), (4096, "2", "0")])
def test_stream_reader_with_tiny_bgzf_blocks(tmp_path, block, threads, chunk):
    """records (and their fixed 36-byte part) straddle many BGZF blocks and superchunks; also an empty BAM"""
    rec, _ = synth.synth_wgs(150, seed=8, n_contigs=2, contig_len=50_000, read_len=151)
    bam = str(tmp_path / "tiny.bam")
    bamio.write_bam(bam, rec, block=block, index=False)
    a = _run(["_dump", bam])
    b = _run(["_dump", bam, "stream", "37"], env=dict(os.environ, STRL_THREADS=threads, STRL_CHUNK_BLOCKS=chunk))
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    assert a.stdout == b.stdout and a.stdout.count("\n") >= rec.n
    empty = synth.synth_wgs(150, seed=8, n_contigs=2, contig_len=50_000)[0]
    from strling_amd.records import RecordBatch
    none = RecordBatch.from_fields([], [], [], [], [], [], [], [], [], targets=empty.targets)
    bam0 = str(tmp_path / "empty.bam")
    bamio.write_bam(bam0, none, index=False)
    c = _run(["_dump", bam0, "stream"], env=dict(os.environ, STRL_THREADS=threads))
    assert c.returncode == 0 and c.stdout == _run(["_dump", bam0]).stdout and c.stdout.count("\n") == 3


def test_indexed_region_reads(tmp_path):
    """.bai linear index + region read == a scan of all records with htslib's iterator filter (tid, pos < end, endpos > beg)"""
    rec, _ = synth.synth_wgs(5000, seed=3, n_contigs=3, contig_len=200_000)
    bam = str(tmp_path / "r.bam")
    bamio.write_bam(bam, rec)
    assert os.path.exists(bam + ".bai")
    stop = np.array([int(rec.pos[i]) + bamio._ref_len(rec, i) for i in range(rec.n)])
    rng = np.random.default_rng(1)
    regions = [(0, 0, 500), (2, 199_000, 200_500), (1, 16_300, 16_400), (1, 150_000, 150_001), (0, 100_000, 140_000), (2, 0, 1)]
    regions += [(int(rng.integers(0, 3)), int(a), int(a) + int(rng.integers(1, 3000))) for a in rng.integers(0, 199_000, 20)]
    for tid, beg, end in regions:
        r = _run(["_region", bam, str(tid), str(beg), str(end)])
        assert r.returncode == 0, r.stderr
        got = [tuple(l.split("\t")) for l in r.stdout.splitlines()]
        sel = np.nonzero((rec.tid == tid) & (rec.pos < end) & (stop > beg))[0]
        exp = [(rec.qname(i).decode(), str(int(rec.pos[i])), str(int(rec.flag[i]))) for i in sel]
        assert got == exp, (tid, beg, end)
    assert sum(1 for t, b, e in regions if np.any((rec.tid == t) & (rec.pos < e) & (stop > b))) > 15


@pytest.mark.parametrize("block", [0xFF00, 2500])
def test_planned_region_reads(tmp_path, block):
    """the host half of `strling call`'s evidence reads on the device -- index span, run of blocks from their headers, the walk
    rule of strl_regions_fetch (zlib standing in for the GPU), the in-memory record parser -- returns htslib's records too"""
    rec, _ = synth.synth_wgs(5000, seed=4, n_contigs=3, contig_len=200_000)
    bam = str(tmp_path / "r.bam")
    bamio.write_bam(bam, rec, block=block, level=6)
    stop = np.array([int(rec.pos[i]) + bamio._ref_len(rec, i) for i in range(rec.n)])
    rng = np.random.default_rng(2)
    regions = [(0, 0, 500), (2, 199_000, 200_500), (1, 16_300, 16_400), (1, 150_000, 150_001), (0, 100_000, 140_000), (2, 0, 1), (1, 16_000, 33_000)]
    regions += [(int(rng.integers(0, 3)), int(a), int(a) + int(rng.integers(1, 3000))) for a in rng.integers(0, 199_000, 25)]
    planned = 0
    for tid, beg, end in regions:
        r = _run(["_region", bam, str(tid), str(beg), str(end), "plan"])
        assert r.returncode == 0, r.stderr
        planned += "plan: " in r.stderr and "host reader" not in r.stderr
        got = [tuple(l.split("\t")) for l in r.stdout.splitlines()]
        sel = np.nonzero((rec.tid == tid) & (rec.pos < end) & (stop > beg))[0]
        exp = [(rec.qname(i).decode(), str(int(rec.pos[i])), str(int(rec.flag[i]))) for i in sel]
        assert got == exp, (tid, beg, end, r.stderr)
    assert planned >= 20


@pytest.mark.gpu
def test_extract_bin_is_byte_identical_to_oracle(sample, oracle):
    """strling extract BAM BIN  ==  the oracle's extract + .bin writer, byte for byte (several GPU batches)."""
    rec, g = sample["rec"], sample["g"]
    out = str(sample["dir"] / "s.bin")
    r = _run(["extract", "-g", sample["bed"], "-v", "--batch", "5000", sample["bam"], out])
    assert r.returncode == 0, r.stderr
    assert "[strling] collecting str-like reads" in r.stderr and "[strling] finished extraction" in r.stderr
    frag = synth.frag_hist(rec)              # < 100k records: the reference falls back to the skipped reads (utils.nim:105-111)
    med = oracle.median(frag)
    exp_t = oracle.extract(rec, g, oracle.make_opts(med, 0.8, 40))
    assert len(exp_t) > 100
    exp = oracle.bin_write(0.8, 40, frag, sample["hdr"].rstrip("\0"), exp_t, rec.qname_off, rec.qnames)
    assert open(out, "rb").read() == exp


@pytest.mark.gpu
def test_extract_device_option_and_both_feeds_write_the_same_bin(sample):
    """--device K / STRL_DEVICE (one `strling` process per sample, each on its own GPU: pipelines/bpipe.config:4): ordinals wrap
    around the devices there are; the compressed bytes copied out of the file's mapping (default) or read with pread
    (STRL_FEED=pread): one .bin"""
    outs = []
    for k, (args, env) in enumerate(((["--device", "0"], {}), (["--device", "5"], {}), ([], {"STRL_DEVICE": "3"}), ([], {"STRL_FEED": "pread"}),
                                     (["--gpus", "2", "--device", "1"], {}))):
        out = str(sample["dir"] / f"dev{k}.bin")
        r = _run(["extract", "-v", "-g", sample["bed"]] + args + [sample["bam"], out], env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        assert "context(s) on device(s)" in r.stderr
        outs.append(open(out, "rb").read())
    assert all(o == outs[0] for o in outs) and len(outs[0]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["crc", "payload"])
def test_extract_reports_a_damaged_bgzf_block(sample, what):
    """a block whose CRC-32 field was changed: the device front end stops like htslib does; a damaged DEFLATE payload: the
    device refuses the block, the host reader (zlib's verdict) takes over and refuses it too"""
    from test_front_device import _corrupt
    bad = str(sample["dir"] / f"bad_{what}.bam")
    _corrupt(sample["bam"], bad, what)
    r = _run(["extract", "-g", sample["bed"], bad, str(sample["dir"] / "bad.bin")])
    assert r.returncode != 0
    # (a flipped payload bit either breaks the DEFLATE stream -- the device refuses the block, the host reader takes over and zlib
    # refuses it too -- or leaves a stream that inflates to other bytes of the same size: then the CRC-32 catches it)
    assert "CRC32" in r.stderr or ("repeating the extraction with the host reader" in r.stderr and "error reading" in r.stderr), r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("shares", ["1", "0"])
@pytest.mark.parametrize("gpus,blocks", [("2", "3"), ("3", "5"), ("2", "8192"), ("4", "2")])
def test_extract_on_several_contexts_writes_the_same_bin(sample, gpus, blocks, shares):
    """strling extract --gpus N, both ways of spreading the file: a contiguous share per context cut at record starts the .bai
    names (every context a feeding thread of its own, nothing carried between contexts) or, STRL_SHARES=0 / no index, the
    file's chunks round-robin over N contexts (partial records carried from one context's chunk to the next context's);
    per-read state gathered on the first, pair logic there: the .bin of the one-GPU run, byte for byte"""
    one = str(sample["dir"] / "one.bin")
    r = _run(["extract", "-g", sample["bed"], sample["bam"], one])
    assert r.returncode == 0, r.stderr
    out = str(sample["dir"] / f"g{gpus}_{blocks}_{shares}.bin")
    r = _run(["extract", "-g", sample["bed"], "-v", "--gpus", gpus, sample["bam"], out], env=dict(os.environ, STRL_CHUNK_BLOCKS=blocks, STRL_SHARES=shares))
    assert r.returncode == 0, r.stderr
    if shares == "1":
        assert f"over {gpus} contexts" in r.stderr and "a contiguous share of the file each" in r.stderr, r.stderr
        assert r.stderr.count("[strling] share ") == int(gpus), r.stderr
    elif int(blocks) < 100:
        assert f"over {gpus} contexts" in r.stderr and "in turn" in r.stderr
    assert open(out, "rb").read() == open(one, "rb").read()


@pytest.fixture(scope="module")
def small_blocks(tmp_path_factory):
    """the same kind of sample in BGZF blocks of 1500 bytes: records straddle blocks, a share holds hundreds of blocks"""
    d = tmp_path_factory.mktemp("cli_sb")
    rec, g = synth.synth_wgs(5000, seed=33, contig_len=600_000)
    bam = str(d / "sb.bam")
    bamio.write_bam(bam, rec, block=1500)
    bed = str(d / "ref.fa.str")
    bamio.write_genome_bed(bed, g, rec.targets)
    one = str(d / "one.bin")
    return dict(dir=d, bam=bam, bed=bed, one=one)


@pytest.mark.gpu
@pytest.mark.parametrize("gpus,blocks", [("2", "64"), ("3", "7"), ("5", "16"), ("8", "3")])
def test_extract_shares_cut_inside_blocks(small_blocks, gpus, blocks):
    """shares whose ends fall INSIDE a BGZF block (the block is inflated by both neighbours; the first stops in front of the
    record the second starts at) and whose chunks end inside records: same .bin"""
    sb = small_blocks
    if not os.path.exists(sb["one"]):
        r = _run(["extract", "-g", sb["bed"], sb["bam"], sb["one"]])
        assert r.returncode == 0, r.stderr
    out = str(sb["dir"] / f"g{gpus}_{blocks}.bin")
    r = _run(["extract", "-g", sb["bed"], "-v", "--gpus", gpus, sb["bam"], out], env=dict(os.environ, STRL_CHUNK_BLOCKS=blocks))
    assert r.returncode == 0, r.stderr
    assert "a contiguous share of the file each" in r.stderr, r.stderr
    assert open(out, "rb").read() == open(sb["one"], "rb").read()


def _shift_bai(src, dst, delta):
    """a .bai whose linear-index offsets point `delta` bytes behind the record starts (a stale / foreign index)"""
    import struct
    b = bytearray(open(src, "rb").read())
    o = 4
    n_ref, = struct.unpack_from("<i", b, o); o += 4
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", b, o); o += 4
        for _ in range(n_bin):
            _, n_chunk = struct.unpack_from("<Ii", b, o); o += 8
            for c in range(n_chunk):
                v, = struct.unpack_from("<Q", b, o)
                struct.pack_into("<Q", b, o, v + delta)
                o += 16
        n_intv, = struct.unpack_from("<i", b, o); o += 4
        for k in range(n_intv):
            v, = struct.unpack_from("<Q", b, o + 8 * k)
            if v:
                struct.pack_into("<Q", b, o + 8 * k, v + delta)
        o += 8 * n_intv
    open(dst, "wb").write(bytes(b))


@pytest.mark.gpu
@pytest.mark.parametrize("delta", [1, 7 << 16])
def test_extract_shares_with_a_lying_index_fall_back_to_chunks(small_blocks, tmp_path, delta):
    """the .bai's record starts are NOT record starts (shifted by a byte) or not even block starts (shifted by 7 file bytes):
    a share does not end where the next begins / its walker finds no BGZF header -- noticed, and the extraction is repeated
    chunk by chunk; the .bin is the one-GPU run's all the same"""
    import shutil
    sb = small_blocks
    if not os.path.exists(sb["one"]):
        r = _run(["extract", "-g", sb["bed"], sb["bam"], sb["one"]])
        assert r.returncode == 0, r.stderr
    bam = str(tmp_path / "lie.bam")
    shutil.copy(sb["bam"], bam)
    _shift_bai(sb["bam"] + ".bai", bam + ".bai", delta)
    out = str(tmp_path / "lie.bin")
    r = _run(["extract", "-g", sb["bed"], "-v", "--gpus", "3", bam, out], env=dict(os.environ, STRL_CHUNK_BLOCKS="16"))
    assert r.returncode == 0, r.stderr
    assert "repeating the extraction chunk by chunk" in r.stderr and "in turn" in r.stderr, r.stderr
    assert open(out, "rb").read() == open(sb["one"], "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("front", ["device", "host"])
def test_extract_never_quits_on_a_qname_carried_by_hundreds_of_records(oracle, tmp_path, front):
    """700 primary records under one qname: more than the device join replays (512 items).  The CLI repeats the extraction
    with the host's string-keyed Cache by itself -- same .bin as the oracle -- instead of quitting (extract.nim never fails there)."""
    from strling_amd.records import RecordBatch
    n = 700
    rec = RecordBatch.from_fields([0] * n, list(range(100, 100 + n)), [0] * n, [5000] * n, [99] * n, [60] * n, ["150M"] * n, ["CAG" * 50] * n, ["dup"] * n,
                                  isize=[300] * n, targets=[("chr1", 100000)])
    bam, bed, out = str(tmp_path / "d.bam"), str(tmp_path / "d.bed"), str(tmp_path / "d.bin")
    hdr = bamio.write_bam(bam, rec, index=False)
    open(bed, "w").close()
    env = dict(os.environ)
    if front == "host":
        env["STRL_FRONT"] = "host"
    r = _run(["extract", "-g", bed, bam, out], env=env)
    assert r.returncode == 0, r.stderr
    assert "repeating the extraction with the host pair logic" in r.stderr
    frag = synth.frag_hist(rec)
    med = oracle.median(frag)
    exp_t = oracle.extract(rec, None, oracle.make_opts(med, 0.8, 40))
    exp = oracle.bin_write(0.8, 40, frag, hdr.rstrip("\0"), exp_t, rec.qname_off, rec.qnames)
    assert open(out, "rb").read() == exp


@pytest.mark.gpu
@pytest.mark.parametrize("front", ["device", "host"])
def test_extract_beyond_the_record_limit_of_a_device_pass(sample, oracle, front):
    """A file with more records than one device pass over a whole input takes (2^31 - 16; here lowered through the test hook
    STRL_RECORD_LIMIT) is not refused -- the reference has no cap (extract.nim:308) -- but goes to the streaming host Cache:
    same .bin as the oracle."""
    rec, g = sample["rec"], sample["g"]
    out = str(sample["dir"] / f"limit_{front}.bin")
    env = dict(os.environ, STRL_RECORD_LIMIT=str(rec.n // 2), STRL_CHUNK_BLOCKS="64")
    if front == "host":
        env["STRL_FRONT"] = "host"
    r = _run(["extract", "-g", sample["bed"], "--batch", "5000", sample["bam"], out], env=env)
    assert r.returncode == 0, r.stderr
    assert "records in one device pass: repeating the extraction with the host pair logic" in r.stderr
    frag = synth.frag_hist(rec)
    exp_t = oracle.extract(rec, g, oracle.make_opts(oracle.median(frag), 0.8, 40))
    exp = oracle.bin_write(0.8, 40, frag, sample["hdr"].rstrip("\0"), exp_t, rec.qname_off, rec.qnames)
    assert open(out, "rb").read() == exp


@pytest.mark.gpu
def test_merge_bounds_match_oracle(sample, oracle, tmp_path):
    """strling merge BIN... -> -bounds.txt identical (rows and row order) to the oracle's merge clustering."""
    bins, all_t, frag_sum = [], [], np.zeros(4096, np.uint64)
    for s in range(3):
        rec, g = synth.synth_wgs(5000, seed=100 + s, contig_len=300_000, str_frac=0.05)
        frag = synth.frag_hist(rec)
        med = oracle.median(frag)
        t = oracle.extract(rec, g, oracle.make_opts(med, 0.8, 40))
        p = str(tmp_path / f"s{s}.bin")
        tt = np.zeros(len(t), api.TREAD_DTYPE)
        for f in tt.dtype.names:
            tt[f] = t[f]
        api.bin_write(p, 0.8, 40, frag, bamio.sam_header(rec.targets), tt, rec.qname_off, rec.qnames)
        bins.append(p)
        keep = t[t["tid"] >= 0].copy()
        keep["qname_id"] = s
        all_t.append(keep)
        frag_sum += frag
    frag_sum = frag_sum.astype(np.uint32)
    prefix = str(tmp_path / "joint")
    r = _run(["merge", "-m", "2", "-o", prefix] + bins)
    assert r.returncode == 0, r.stderr
    window = oracle.median(frag_sum, 0.98)
    mcd = int(0.5 * oracle.median(frag_sum, 0.5))
    exp_b, _ = oracle.call_bounds(np.concatenate(all_t), 0, window, min_support=2, max_clip_dist=mcd)
    exp = ["#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total"]
    exp += [oracle.bounds_row(b, rec.targets[int(b["tid"])][0]) for b in exp_b]
    got = open(prefix + "-bounds.txt").read().rstrip("\n").split("\n")
    assert len(exp) > 3
    assert got == exp
    # ... and clustered on several contexts (shares of the reads, exchange, owned groups, the reference's row order): the same file
    for gpus in ("2", "3"):
        pg = str(tmp_path / f"joint{gpus}")
        r = _run(["merge", "-m", "2", "-v", "--gpus", gpus, "-o", pg] + bins)
        assert r.returncode == 0, r.stderr
        assert f"clustered on {gpus} contexts" in r.stderr
        assert open(pg + "-bounds.txt").read() == open(prefix + "-bounds.txt").read(), gpus


def _write_fasta(path, contigs, width=70, gz=False):
    import gzip
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for name, seq in contigs:
            f.write(b">" + name.encode() + b" some description\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + b"\n")


@pytest.mark.gpu
@pytest.mark.parametrize("gz", [False, True])
def test_index_bed_is_identical_to_oracle(oracle, tmp_path, gz):
    """strling index FASTA -> <FASTA>.str rows == the oracle's repeat_windows/trim rows (genome_strs.nim:61-137)"""
    contigs = [("chr1", synth.synth_chrom(300_000, 31)), ("chrEmpty", b""), ("chr2", synth.synth_chrom(70_001, 32)), ("tiny", b"ACGTACGTAC")]
    fa = str(tmp_path / ("ref.fa.gz" if gz else "ref.fa"))
    _write_fasta(fa, contigs, gz=gz)
    out = str(tmp_path / "ref.str")
    r = _run(["index", "-g", out, fa])
    assert r.returncode == 0, r.stderr
    assert f"Writing genome str index to: {out}" in r.stderr and "STR-like regions in the genome" in r.stderr
    exp = "".join(f"{name}\t{a}\t{b}\t{u}\n" for name, seq in contigs for a, b, u in oracle.index_chrom(seq.upper(), 0.8))
    assert exp.count("\n") > 50
    assert open(out).read() == exp
    # an existing file is left alone (genome_strs.nim:139-140), default output name is ./<FASTA>.str
    open(out, "w").write("untouched\n")
    r = _run(["index", "-g", out, "-p", "0.7", fa])
    assert r.returncode == 0 and "using existing file" in r.stderr and open(out).read() == "untouched\n"
    r = _run(["index", "-p", "0.7", fa], cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    exp7 = "".join(f"{name}\t{a}\t{b}\t{u}\n" for name, seq in contigs for a, b, u in oracle.index_chrom(seq.upper(), 0.7))
    assert open(str(tmp_path / (os.path.basename(fa) + ".str"))).read() == exp7


@pytest.mark.gpu
def test_extract_builds_missing_genome_index(oracle, tmp_path):
    """extract -f FASTA -g MISSING creates the index first (genome_strs.nim:124-138) and then uses it."""
    n_contigs, clen = 3, 200_000
    contigs = [(f"chr{i + 1}", synth.synth_chrom(clen, 40 + i)) for i in range(n_contigs)]
    fa = str(tmp_path / "ref.fa")
    _write_fasta(fa, contigs)
    rec, _ = synth.synth_wgs(4000, seed=77, n_contigs=n_contigs, contig_len=clen)
    bam = str(tmp_path / "s.bam")
    hdr = bamio.write_bam(bam, rec)
    bed = str(tmp_path / "made.str")
    out = str(tmp_path / "s.bin")
    r = _run(["extract", "-f", fa, "-g", bed, bam, out])
    assert r.returncode == 0, r.stderr
    rows = [(name, a, b) for name, seq in contigs for a, b, u in oracle.index_chrom(seq.upper(), 0.8)]
    assert [tuple(l.split("\t")[:3]) for l in open(bed).read().splitlines()] == [(n, str(a), str(b)) for n, a, b in rows]
    from strling_amd.records import GenomeStr
    per = {i: [(a, b) for n, a, b in rows if n == name] for i, (name, _) in enumerate(contigs) if any(n == name for n, _, _ in rows)}
    g = GenomeStr.from_lists(n_contigs, per)
    frag = synth.frag_hist(rec)
    exp_t = oracle.extract(rec, g, oracle.make_opts(oracle.median(frag), 0.8, 40))
    exp = oracle.bin_write(0.8, 40, frag, hdr.rstrip("\0"), exp_t, rec.qname_off, rec.qnames)
    assert open(out, "rb").read() == exp
    # without -g the index goes to a temporary file that is removed again; without -f and -g it cannot be built
    out2 = str(tmp_path / "s2.bin")
    r = _run(["extract", "-f", fa, bam, out2])
    assert r.returncode == 0, r.stderr
    assert open(out2, "rb").read() == exp
    r = _run(["extract", bam, out2])
    assert r.returncode == 1 and "couldn't open fasta" in r.stderr


def _three_bins(oracle, tmp_path, targets_of=None):
    bins, per_sample, frag_sum, rec = [], [], np.zeros(4096, np.uint64), None
    for s in range(3):
        rec, g = synth.synth_wgs(5000, seed=200 + s, contig_len=300_000, str_frac=0.05)
        frag = synth.frag_hist(rec)
        t = oracle.extract(rec, g, oracle.make_opts(oracle.median(frag), 0.8, 40))
        p = str(tmp_path / f"m{s}.bin")
        tt = np.zeros(len(t), api.TREAD_DTYPE)
        for f in tt.dtype.names:
            tt[f] = t[f]
        targets = targets_of(s, rec.targets) if targets_of else rec.targets
        api.bin_write(p, 0.8, 40, frag, bamio.sam_header(targets), tt, rec.qname_off, rec.qnames)
        bins.append(p)
        keep = t[t["tid"] >= 0].copy()
        keep["qname_id"] = s
        per_sample.append(keep)
        frag_sum += frag
    return bins, per_sample, frag_sum.astype(np.uint32), rec.targets


def _write_fai(path, targets):
    open(path, "w").write(">x\nA\n")
    with open(path + ".fai", "w") as f:
        off = 0
        for name, ln in targets:
            f.write(f"{name}\t{ln}\t{off}\t70\t71\n")
            off += ln + ln // 70 + 10


@pytest.mark.gpu
def test_merge_chromosome_restricts_reads_and_loci(oracle, tmp_path):
    """merge --chromosome (merge.nim:52,89,101; unpack.nim:126; cluster.nim:139): the by-chromosome joint pipeline"""
    bins, per_sample, frag, targets = _three_bins(oracle, tmp_path)
    fa = str(tmp_path / "ref.fa")
    _write_fai(fa, targets)
    window, mcd = oracle.median(frag, 0.98), int(0.5 * oracle.median(frag, 0.5))
    header = "#chrom\tleft\tright\trepeat\tname\tleft_most\tright_most\tcenter_mass\tn_left\tn_right\tn_total"
    allt = np.concatenate(per_sample)
    seen_rows = 0
    for tid in (0, 7, len(targets) - 1):
        prefix = str(tmp_path / f"chrom{tid}")
        r = _run(["merge", "-m", "2", "-f", fa, "--chromosome", targets[tid][0], "-o", prefix] + bins)
        assert r.returncode == 0, r.stderr
        exp_b, _ = oracle.call_bounds(allt[allt["tid"] == tid], 0, window, min_support=2, max_clip_dist=mcd)
        exp = [header] + [oracle.bounds_row(b, targets[int(b["tid"])][0]) for b in exp_b]
        assert open(prefix + "-bounds.txt").read().rstrip("\n").split("\n") == exp
        seen_rows += len(exp_b)
    assert seen_rows > 0
    # a -l BED keeps only the loci of that chromosome
    bed = str(tmp_path / "loci.bed")
    open(bed, "w").write(f"{targets[0][0]}\t1000\t1020\tCAG\tl0\n{targets[7][0]}\t2000\t2030\tAT\n")
    prefix = str(tmp_path / "chrom_bed")
    r = _run(["merge", "-m", "2", "-f", fa, "--chromosome", targets[7][0], "-l", bed, "-o", prefix] + bins)
    assert r.returncode == 0, r.stderr
    rows = open(prefix + "-bounds.txt").read().rstrip("\n").split("\n")
    assert rows[1].split("\t")[:4] == [targets[7][0], "2000", "2030", "AT"] and all(x.split("\t")[0] == targets[7][0] for x in rows[1:])
    # errors of get_tid(fasta, chromosome), merge.nim:36-45
    r = _run(["merge", "-f", fa, "--chromosome", "nope", "-o", prefix] + bins)
    assert r.returncode == 1 and "chromosome: nope not found in fasta" in r.stderr
    r = _run(["merge", "-f", str(tmp_path / "missing.fa"), "--chromosome", "chr1", "-o", prefix] + bins)
    assert r.returncode == 1 and "could not open fasta" in r.stderr


@pytest.mark.gpu
def test_merge_diff_refs(oracle, tmp_path):
    """-d lets bins with differing headers through (merge.nim:60,107-108); chromosome names then come from the first bin, or
    from the .fai of -f when both are given (merge.nim:84-86)"""
    def targets_of(s, targets):
        return targets if s != 1 else [(n + "_alt", ln) for n, ln in targets]
    bins, per_sample, frag, targets = _three_bins(oracle, tmp_path, targets_of)
    prefix = str(tmp_path / "d")
    r = _run(["merge", "-m", "2", "-o", prefix] + bins)
    assert r.returncode == 1 and "inconsistent bam header" in r.stderr
    r = _run(["merge", "-m", "2", "-d", "-o", prefix] + bins)
    assert r.returncode == 0, r.stderr
    window, mcd = oracle.median(frag, 0.98), int(0.5 * oracle.median(frag, 0.5))
    exp_b, _ = oracle.call_bounds(np.concatenate(per_sample), 0, window, min_support=2, max_clip_dist=mcd)
    got = open(prefix + "-bounds.txt").read().rstrip("\n").split("\n")[1:]
    assert got == [oracle.bounds_row(b, targets[int(b["tid"])][0]) for b in exp_b] and len(got) > 3
    fa = str(tmp_path / "other.fa")
    renamed = [("fa_" + n, ln) for n, ln in targets]
    _write_fai(fa, renamed)
    r = _run(["merge", "-m", "2", "-d", "-f", fa, "-o", prefix] + bins)
    assert r.returncode == 0, r.stderr
    got = open(prefix + "-bounds.txt").read().rstrip("\n").split("\n")[1:]
    assert got == [oracle.bounds_row(b, renamed[int(b["tid"])][0]) for b in exp_b]


@pytest.mark.gpu
def test_damaged_cram_input_is_refused_with_a_reason(tmp_path):
    """a file that says CRAM but holds nothing readable: exit code 1 and the reason (tests/test_cram.py covers real CRAM input)"""
    p = str(tmp_path / "x.cram")
    open(p, "wb").write(b"CRAM\x03\x00" + b"\0" * 64)
    r = _run(["extract", p, str(tmp_path / "x.bin")])
    assert r.returncode == 1 and "x.cram: CRAM container header CRC32 mismatch" in r.stderr      # (every byte of a CRAM lies under a checksum; htslib checks them too)


def test_malformed_bgzf_blocks_are_rejected(tmp_path):
    """a BC size smaller than the block's own header and an extra subfield running past XLEN: an error, not an allocation
    of gigabytes / a read past the buffer (both readers)"""
    import struct
    rec, _ = synth.synth_wgs(50, seed=3, contig_len=100_000)
    good = str(tmp_path / "g.bam")
    bamio.write_bam(good, rec)
    data = bytearray(open(good, "rb").read())
    bad1 = bytes(data[:16]) + struct.pack("<H", 5) + bytes(data[18:])            # BSIZE - 1 = 5 < header size
    bad2 = bytes(data[:14]) + struct.pack("<H", 200) + bytes(data[16:])          # subfield length 200 > XLEN
    for i, b in enumerate((bad1, bad2)):
        p = str(tmp_path / f"bad{i}.bam")
        open(p, "wb").write(b)
        for mode in ([], ["stream"]):
            r = _run(["_dump", p] + mode)
            assert r.returncode != 0 and ("BGZF" in r.stderr or "couldn't open" in r.stderr), r.stderr[-300:]


def _fnv_records(rec):
    """the checksum `strling _decode` prints, computed from the batch the BAM was written from: FNV-1a over every field of
    every record in file order (tid, pos, mtid, mpos, isize, flag, mapq, l_seq, cigar words, SEQ bytes, qname bytes)"""
    M = (1 << 64) - 1
    s = 0xcbf29ce484222325
    P = 0x100000001b3
    u32 = lambda v: int(v) & 0xffffffff
    seq4 = rec.seq4
    qn = bytes(rec.qnames)
    for i in range(rec.n):
        vals = [u32(rec.tid[i]), u32(rec.pos[i]), u32(rec.mtid[i]), u32(rec.mpos[i]), u32(rec.isize[i]), int(rec.flag[i]), int(rec.mapq[i]), u32(rec.l_seq[i])]
        vals += [int(c) for c in rec.cigar[int(rec.cigar_off[i]):int(rec.cigar_off[i + 1])]]
        o = int(rec.seq_off[i])
        vals += seq4[o:o + (int(rec.l_seq[i]) + 1) // 2].tolist()
        vals += list(qn[int(rec.qname_off[i]):int(rec.qname_off[i + 1])])
        for v in vals:
            s = ((s ^ v) * P) & M
    return s


def test_stream_reader_is_the_same_for_every_engine_and_shape(tmp_path):
    """CPU only (`strling _decode` needs no GPU): the multi-threaded reader -- header-walk thread, background superchunk
    loads, speculative record starts, own DEFLATE decoder or zlib -- delivers exactly the records that were written, in
    order, whatever the thread count, superchunk size, batch size or inflate engine."""
    rec, _ = synth.synth_wgs(3000, seed=77, contig_len=300_000, n_contigs=3)
    bam = str(tmp_path / "d.bam")
    bamio.write_bam(bam, rec, level=6)
    want = _fnv_records(rec)
    for env, batch in (({"STRL_THREADS": "1"}, "1048576"), ({"STRL_THREADS": "4"}, "1000"), ({"STRL_THREADS": "3", "STRL_CHUNK_BLOCKS": "1"}, "777"),
                       ({"STRL_THREADS": "4", "STRL_CHUNK_BLOCKS": "2", "STRL_INFLATE": "zlib"}, "65536"), ({"STRL_THREADS": "7", "STRL_CHUNK_BLOCKS": "3"}, "5")):
        r = _run(["_decode", bam, batch], env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        line = [l for l in r.stderr.splitlines() if "decoded" in l][-1]
        assert f"decoded {rec.n} records" in line, line
        assert f"(checksum {want})" in line, (env, line)


def _bam_layout(path):
    """(block file offset, inflated offset, inflated size) of every data block, the inflated bytes, every record start in them"""
    import struct
    import zlib
    raw = open(path, "rb").read()
    blocks, infl, o = [], bytearray(), 0
    while o < len(raw):
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        data = zlib.decompress(raw[o + 12 + xlen:o + bsize - 8], -15)
        blocks.append((o, len(infl), len(data)))
        infl += data
        o += bsize
    l_text, = struct.unpack_from("<i", infl, 4)
    q = 8 + l_text
    n_ref, = struct.unpack_from("<i", infl, q)
    q += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", infl, q)
        q += 8 + l_name
    starts = set()
    while q < len(infl):
        starts.add(q)
        q += 4 + struct.unpack_from("<i", infl, q)[0]
    return blocks, starts


@pytest.mark.parametrize("G,max_blocks", [(2, 4096), (3, 5), (7, 64), (16, 1)])
def test_shares_tile_the_file_at_record_starts(tmp_path, G, max_blocks):
    """`extract --gpus G` cuts the file at record starts the .bai names: every cut IS a record start (checked against a walk of
    the inflated file), the shares' blocks tile the file's data blocks with the block a cut falls into delivered to both
    neighbours, and the bytes a share leaves to the next add up (CPU: the header walkers alone, `strling _shares`)"""
    rec, g = synth.synth_wgs(3000, seed=5, contig_len=300_000)
    bam = str(tmp_path / "t.bam")
    bamio.write_bam(bam, rec, block=1200)
    blocks, starts = _bam_layout(bam)
    by_off = {b[0]: b for b in blocks}
    r = _run(["_shares", bam, str(G), str(max_blocks)])
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in r.stdout.strip().split("\n")]
    n = int(rows[0][1])
    assert 2 <= n <= G and len(rows) == n + 1
    total = 0
    prev_end_uoff = None
    for k, row in enumerate(rows[1:]):
        assert row[0] == "share" and int(row[1]) == k and row[2] != "error", row
        coff, uoff, nb, isz, trim, runs, first_c, last_end, saw_last = (int(x) for x in row[2:])
        assert saw_last == 1
        b = by_off[coff]
        assert b[1] + uoff in starts, "a cut that is not a record start"
        assert first_c == coff + 18                      # the share's first block is the one the cut names
        if prev_end_uoff is not None:
            assert uoff == prev_end_uoff                  # where the previous share stopped inside the shared block
        own = isz - uoff - trim
        total += own
        if k + 1 < n:
            ncoff, nuoff = int(rows[k + 2][2]), int(rows[k + 2][3])
            nb_ = by_off[ncoff]
            assert b[1] + uoff + own == nb_[1] + nuoff    # the share ends exactly at the next cut
            assert trim == (nb_[2] - nuoff if nuoff else 0)
            prev_end_uoff = nuoff
        else:
            assert trim == 0
    first = by_off[int(rows[1][2])]
    assert first[1] + int(rows[1][3]) + total == blocks[-1][1] + blocks[-1][2]   # first record .. end of the data


@pytest.mark.gpu
@pytest.mark.parametrize("limit_mb,hint", [("700", "10000000")])
def test_extract_out_of_device_memory_takes_the_host_pair_logic(sample, limit_mb, hint):
    """the per-read state of a whole input is resident on the device (~130 B per read: a 288 GB device holds ~2e9 reads); past
    that -- simulated here with STRL_DEVICE_MEM_LIMIT_MB: 700 MB against the state of 1e7 reads (the host pair logic's own
    batches take ~100 MB) -- the CLI says what ran out and repeats the extraction with the streaming host Cache, which keeps nothing per read
    on the device: same .bin, no HIP error text, exit code 0"""
    one = str(sample["dir"] / "one.bin")
    r = _run(["extract", "-g", sample["bed"], sample["bam"], one])
    assert r.returncode == 0, r.stderr
    out = str(sample["dir"] / f"nomem{limit_mb}.bin")
    env = dict(os.environ, STRL_DEVICE_MEM_LIMIT_MB=limit_mb, STRL_READS_HINT=hint, STRL_CHUNK_BLOCKS="4")
    r = _run(["extract", "-g", sample["bed"], sample["bam"], out], env=env)
    assert r.returncode == 0, r.stderr
    assert "out of device memory" in r.stderr and "repeating the extraction with the host pair logic" in r.stderr, r.stderr
    assert open(out, "rb").read() == open(one, "rb").read()


@pytest.mark.gpu
def test_extract_gpus_without_an_index_and_on_a_tiny_file(sample, tmp_path):
    """--gpus N where shares cannot be cut: a BAM without a .bai goes over the contexts chunk by chunk (`in turn`); a file of one
    block has no second record start in its index worth a share -- both give the one-GPU .bin"""
    import shutil
    one = str(sample["dir"] / "one.bin")
    r = _run(["extract", "-g", sample["bed"], sample["bam"], one])
    assert r.returncode == 0, r.stderr
    bam = str(tmp_path / "noindex.bam")
    shutil.copy(sample["bam"], bam)
    out = str(tmp_path / "noindex.bin")
    r = _run(["extract", "-g", sample["bed"], "-v", "--gpus", "3", bam, out], env=dict(os.environ, STRL_CHUNK_BLOCKS="4"))
    assert r.returncode == 0, r.stderr
    assert "no .bai record starts to cut the file at" in r.stderr and "in turn" in r.stderr, r.stderr
    assert open(out, "rb").read() == open(one, "rb").read()
    rec, g = synth.synth_wgs(60, seed=3, contig_len=50_000)
    tiny, bed = str(tmp_path / "tiny.bam"), str(tmp_path / "tiny.str")
    bamio.write_bam(tiny, rec)
    bamio.write_genome_bed(bed, g, rec.targets)
    r1 = _run(["extract", "-g", bed, tiny, str(tmp_path / "t1.bin")])
    r4 = _run(["extract", "-g", bed, "-v", "--gpus", "4", tiny, str(tmp_path / "t4.bin")])
    assert r1.returncode == 0 and r4.returncode == 0, r4.stderr
    assert open(str(tmp_path / "t4.bin"), "rb").read() == open(str(tmp_path / "t1.bin"), "rb").read()

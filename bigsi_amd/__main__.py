"""`python -m bigsi_amd <command>`: the query-side commands of the reference CLI (bigsi/__main__.py:103-320) on the
hip-hbm backend.  argparse instead of hug; same command names, arguments and output text."""
import argparse
import json
import os
import sys

import yaml

from . import BIGSI
from .bitrow import BitRow
from .frontend import bulk_search, read_fasta, search, variant_search
from .graph.bigsi import DEFAULT_CONFIG
from .storage import get_storage
from .utils import seq_to_kmers


def get_config_from_file(config_file):            # __main__.py:86-94
    config_file = config_file or os.environ.get("BIGSI_CONFIG")
    if not config_file:
        return DEFAULT_CONFIG
    with open(config_file) as f:
        return yaml.safe_load(f)


def main(argv=None):
    p = argparse.ArgumentParser(prog="bigsi_amd")
    sub = p.add_subparsers(dest="cmd", required=True)

    def common(sp):
        sp.add_argument("--config", "-c", default=None)
        return sp

    def shardable(sp):
        sp.add_argument("--sharded", action="store_true",
                        help="the index is spread by column range over the ranks of a `python -m torch.distributed.run` launch "
                             "(one process per GPU); rank 0 prints")
        sp.add_argument("--out", "-o", default=None,
                        help="with --sharded: write the text to this file instead of stdout (transport libraries announce "
                             "themselves on stdout)")
        return sp

    sp = shardable(common(sub.add_parser("search")))
    sp.add_argument("seq")
    sp.add_argument("--threshold", "-t", type=float, default=1.0)
    sp.add_argument("--score", action="store_true")
    sp.add_argument("--format", choices=["json", "csv"], default="json")
    sp = shardable(common(sub.add_parser("bulk_search")))
    sp.add_argument("fasta")
    sp.add_argument("--threshold", "-t", type=float, default=1.0)
    sp.add_argument("--score", action="store_true")
    sp.add_argument("--format", choices=["json", "csv"], default="json")
    sp.add_argument("--stream", action="store_true")
    sp = common(sub.add_parser("variant_search"))
    sp.add_argument("reference")
    sp.add_argument("ref")
    sp.add_argument("pos", type=int)
    sp.add_argument("alt")
    sp.add_argument("--gene", "-g", default=None)
    sp.add_argument("--genbank", "-b", default=None)
    sp.add_argument("--format", choices=["json", "csv"], default="json")
    sp.add_argument("--probes", default=None, help="output of `mykrobe variants make-probes` for this variant (skips calling it)")
    sp = common(sub.add_parser("bloom", help="Bloom filter of the k-mers of a Cortex .ctx graph, a FASTA file or a one-k-mer-per-line text file"))
    sp.add_argument("infile")
    sp.add_argument("outfile")
    sp = shardable(common(sub.add_parser("build")))
    sp.add_argument("--bloomfilters", "-b", action="append", default=[])
    sp.add_argument("--samples", "-s", action="append", default=[])
    sp.add_argument("--from_file", default=None, help="TSV of bloomfilter path <tab> sample name (bigsi/__main__.py:139-156)")
    sp = common(sub.add_parser("merge", help="append the samples of the index described by MERGE_CONFIG (bigsi/__main__.py:173-181)"))
    sp.add_argument("merge_config")
    sp = common(sub.add_parser("insert"))
    sp.add_argument("bloomfilter")
    sp.add_argument("sample")
    common(sub.add_parser("delete"))
    sp = common(sub.add_parser("hold", help="load the index and keep it resident in HBM for OTHER processes: writes an attach file (hipIpc handle + "
                                            "metadata); `search` / `bulk_search` with storage-config {attach: FILE} then open it in milliseconds"))
    sp.add_argument("--handle", default=None, help="the attach file (default: storage-config `export`, else <filename>.attach)")
    sp.add_argument("--seconds", type=float, default=None, help="exit after this long (default: until SIGTERM / SIGINT)")
    sp.add_argument("--until-eof", action="store_true", help="also exit when stdin ends (a parent process that holds the other end of a pipe)")
    sp = common(sub.add_parser("import-bdb", help="load an existing BerkeleyDB index (v0.3 file, or a v0.1 directory with graph + metadata) into HBM"))
    sp.add_argument("path")
    a = p.parse_args(argv)
    config = get_config_from_file(a.config)

    if getattr(a, "sharded", False):
        return sharded_main(a, config)
    if a.cmd == "search":
        print(search(BIGSI(config), a.seq, a.threshold, a.score, a.format))
    elif a.cmd == "bulk_search":
        text = bulk_search(BIGSI(config), a.fasta, a.threshold, a.score, a.format, a.stream)
        if text is not None:
            print(text)
    elif a.cmd == "variant_search":
        print(variant_search(BIGSI(config), a.reference, a.ref, a.pos, a.alt, a.gene, a.genbank, a.format, a.probes))
    elif a.cmd == "bloom":
        first = open(a.infile, "rb").read(6)
        if first == b"CORTEX":                      # the reference's input: a Cortex graph (bigsi/__main__.py:120-131)
            from .cortex import extract_kmers_from_ctx
            kmers = list(extract_kmers_from_ctx(a.infile, config["k"]))
        elif first[:1] == b">":
            kmers = [km for _, s in read_fasta(a.infile) for km in seq_to_kmers(s, config["k"])]
        else:
            kmers = [l.strip() for l in open(a.infile) if l.strip()]
        with open(a.outfile, "wb") as f:
            f.write(BIGSI.bloom(config, kmers).tobytes())
    elif a.cmd == "build":
        paths, samples = build_inputs(a)
        build_in_slabs(config, paths, samples)
        print('{"result": "success"}')
    elif a.cmd == "merge":
        other_config = get_config_from_file(a.merge_config)
        index = BIGSI(config)
        index.merge(BIGSI(other_config))
        index.storage.sync()                        # the snapshot file is the only thing the next process sees
        print(json.dumps({"result": "merged %s into %s." % (a.merge_config, a.config)}))
    elif a.cmd == "insert":
        index = BIGSI(config)
        index.insert(BitRow.frombytes(open(a.bloomfilter, "rb").read(), config["m"]), a.sample)
        index.storage.sync()                        # persist: the reference's backends write through
        print('{"result": "success"}')
    elif a.cmd == "import-bdb":
        from . import bdb
        dst = get_storage(config)
        if os.path.isdir(a.path):
            print("rows=%d cols=%d k=%d" % bdb.import_v01_index(a.path, dst))
        else:
            print("rows=%d cols=%d" % bdb.import_index(a.path, dst))
    elif a.cmd == "delete":
        get_storage(config).delete_all()
    elif a.cmd == "hold":
        hold(config, a.handle, a.seconds, a.until_eof)
    return 0


def hold(config, handle=None, seconds=None, until_eof=False):
    """The process that keeps an index resident for others (the reference's store is a file every request and pool worker opens
    again, bigsi/__main__.py:75-80, 204-205; a 125 GB matrix should be ingested once).  Prints one line when the attach file is
    in place, then waits -- for SIGTERM / SIGINT, `seconds`, or (until_eof) the end of stdin -- removes the file and exits."""
    import signal
    import threading
    import time
    index = BIGSI(config)
    sc = config.get("storage-config", {})
    path = handle or sc.get("export") or ((sc.get("filename") or "bigsi-%s" % sc.get("name", "default")) + ".attach")
    index.storage.export_attach(path)
    print(json.dumps({"result": "holding", "attach": os.path.abspath(path), "pid": os.getpid(), "num_samples": index.num_samples}), flush=True)
    stop = threading.Event()
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, lambda *_: stop.set())

    def watch_stdin():
        try:
            while sys.stdin.read(4096):
                pass
        except (OSError, ValueError):
            return
        stop.set()
    if until_eof:                                       # (only then: a daemon's stdin is /dev/null, which is at its end at once)
        threading.Thread(target=watch_stdin, daemon=True).start()
    t0 = time.time()
    while not stop.wait(0.2):
        if seconds is not None and time.time() - t0 >= seconds:
            break
    try:
        os.remove(path)
    except OSError:
        pass


def build_inputs(a):
    """(bloom filter paths, sample names) of a `build` command line, as the reference resolves them
    (bigsi/__main__.py:139-160): --from_file XOR -b; sample names default to the filter paths."""
    import csv
    paths, samples = list(a.bloomfilters), list(a.samples)
    if a.from_file and paths:
        raise ValueError("You can only specify blooms via from_file or bloomfilters, but not both")
    if a.from_file:
        paths, samples = [], []
        with open(a.from_file, "r") as tsv:
            for row in csv.reader(tsv, delimiter="\t"):
                paths.append(row[0])
                samples.append(row[1])
    if samples:
        assert len(samples) == len(paths)
    else:
        samples = list(paths)
    return paths, samples


def parse_size(text):
    """'4GB' / '512 MiB' / 1000 -> bytes (the reference hands config['max_build_mem_bytes'] to humanfriendly.parse_size:
    decimal multiples for kB/MB/GB, binary ones for KiB/MiB/GiB)."""
    import re
    if isinstance(text, (int, float)):
        return int(text)
    m = re.fullmatch(r"\s*([0-9.]+)\s*(?:([kmgtp])(i?)b?|b|bytes?)?\s*", str(text).lower())
    if not m:
        raise ValueError("cannot parse size %r" % (text,))
    value = float(m.group(1))
    if m.group(2):
        value *= (1024 if m.group(3) else 1000) ** ("kmgtp".index(m.group(2)) + 1)
    return int(value)


def build_in_slabs(config, paths, samples):
    """BIGSI.build with bounded host memory: the filters are read and sent to the device `max_build_mem_bytes` at a time
    (config key, as in the reference's chunked build, bigsi/cmds/build.py:43-73) -- the first slab builds the index, later slabs
    append their columns in place (the device transpose writes columns [col0, col0+n) of the resident matrix), so no
    temporary indexes and no merges are needed."""
    limit = parse_size(config["max_build_mem_bytes"]) if config.get("max_build_mem_bytes") else None
    per = (int(config["m"]) + 7) // 8
    slab = len(paths) if not limit else max(1, limit // max(per, 1))
    if limit and limit < per:
        raise ValueError("Max memory must be at least the Bloomfilter size in bytes")
    load = lambda b: BitRow.frombytes(open(b, "rb").read(), config["m"])      # noqa: E731
    index = BIGSI.build(config, [load(b) for b in paths[:slab]], samples[:slab])
    for i in range(slab, len(paths), slab):
        index.insert_many([load(b) for b in paths[i:i + slab]], samples[i:i + slab])
    index.storage.sync()
    return index


def sharded_main(a, config):
    """search / bulk_search / build on a column-sharded index: every rank runs the same command (SPMD), the device work
    and the exchange are bigsi_amd.parallel's, and only rank 0 writes to stdout."""
    import io

    from .parallel import ShardedBIGSI
    rank, world, _ = ShardedBIGSI.launch()
    if a.cmd == "build":
        paths, samples = build_inputs(a)
        sb = ShardedBIGSI.build(config, paths, samples)
        text = '{"result": "success"}'
    else:
        sb = ShardedBIGSI.open(config)
        if a.cmd == "search":
            text = search(sb, a.seq, a.threshold, a.score, a.format)
        else:
            sink = io.StringIO()
            text = bulk_search(sb, a.fasta, a.threshold, a.score, a.format, a.stream, out=sink)
            if text is None:                      # --stream: the records were printed into `sink`
                text = sink.getvalue()[:-1]
    if rank == 0:
        if a.out:
            with open(a.out, "w", newline="") as f:
                f.write(text + "\n")
        else:
            print(text)
    sb.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

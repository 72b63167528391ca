#!/usr/bin/env python3
"""EPIK-compatible launcher for the MI355X placement engine.

The command line is the reference launcher's (reference epik.py:29-59): a `place`
command taking -i/--database, -s/--states {nucl,amino}, --omega (1.5), --mu (1.0),
-o/--outputdir, --threads (1), --max-ram and one FASTA file.  Like the reference
(epik.py:73-98) it only selects the native driver -- `epik-dna` for nucl, `epik-aa` for
amino -- translates the options into that driver's flags (-d -q -j --omega --mu -o
[--max-ram]) and runs it.  One option is new: --gpus, the number of MI355X devices the
reads are sharded across.
"""
from __future__ import annotations

import os
import subprocess
import sys

import click

__version__ = "0.2.0"

HERE = os.path.dirname(os.path.realpath(__file__))
DRIVERS = {"nucl": "epik-dna", "amino": "epik-aa"}

# (flags, click keyword arguments) -- one row per option of `place`
PLACE_OPTIONS = [
    (("-i", "--database"), dict(required=True, type=click.Path(exists=True, dir_okay=False),
                                help="Phylo-k-mer database to place against.")),
    (("-s", "--states"), dict(type=click.Choice(sorted(DRIVERS, reverse=True)), default="nucl",
                              show_default=True, help="nucl for DNA, amino for proteins.")),
    (("--omega",), dict(type=float, default=1.5, show_default=True,
                        help="Score-threshold parameter; may exceed the one the database was built with.")),
    (("--mu",), dict(type=float, default=1.0, show_default=True,
                     help="Proportion of the database to load, in (0, 1].")),
    (("-o", "--outputdir"), dict(required=True, type=click.Path(exists=True, file_okay=False),
                                 help="Directory that receives placements_<input>.jplace.")),
    (("--threads",), dict(type=int, default=1, show_default=True,
                          help="Accepted for compatibility; placement runs on the GPU.")),
    (("--max-ram",), dict(type=str, default="", help="Approximate limit on the loaded database, e.g. 512, 256K, 42M, 4.2G.")),
    (("--gpus",), dict(type=int, default=1, show_default=True, help="MI355X devices to shard the reads across.")),
    (("--db-shard",), dict(type=int, default=1, show_default=True,
                           help="Cut the database in this many shards by k-mer code, one per device (a database "
                                "larger than one device's memory); 1 = the whole database on every device.")),
]


def driver_path(states: str) -> str:
    """The native driver: next to this script when installed, else the in-tree build."""
    name = DRIVERS[states]
    for folder in (HERE, os.path.join(HERE, "epik_amd", "bin")):
        candidate = os.path.join(folder, name)
        if os.path.exists(candidate):
            return candidate
    return os.path.join(HERE, "epik_amd", "bin", name)


def driver_command(database, states, omega, mu, outputdir, threads, max_ram, gpus, input_file, db_shard=1):
    argv = [driver_path(states), "-d", str(database), "-q", str(input_file), "-j", str(threads),
            "--omega", str(omega), "--mu", str(mu), "-o", str(outputdir)]
    if max_ram:
        argv += ["--max-ram", max_ram]
    if gpus != 1:
        argv += ["--gpus", str(gpus)]
    if db_shard != 1:
        argv += ["--db-shard", str(db_shard)]
    return argv + [str(input_file)]  # the reference passes the query a second time, positionally


@click.group()
@click.version_option(__version__)
def epik():
    """Phylogenetic placement with informative k-mers on AMD Instinct MI355X."""


def _place(input_file, **options):
    """Places the sequences of a FASTA file:  epik.py place -i DB -o OUTDIR [-s nucl|amino] QUERY.fasta"""
    argv = driver_command(input_file=input_file, **options)
    print(" ".join(argv))
    sys.exit(subprocess.call(argv))


place = click.argument("input_file", type=click.Path(exists=True))(_place)
for flags, kwargs in reversed(PLACE_OPTIONS):
    place = click.option(*flags, **kwargs)(place)
place = epik.command(name="place")(place)


if __name__ == "__main__":
    epik()

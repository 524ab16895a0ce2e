#!/usr/bin/env python3
"""CLI of the preprocessing step, flags as the reference's preprocess.py:21-29:

    python preprocess.py --raw_dataset raw.csv --output data/ml-1m.txt --type movielens [--limit N]
"""
import argparse
import logging

import castrec_amd  # noqa: F401
from castrec_amd.data_reader import main


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)-5.5s]  %(message)s")
    ap = argparse.ArgumentParser()
    ap.add_argument("--raw_dataset", help="raw dump: gzip of review dicts (amazon) or user,item,rating,ts csv")
    ap.add_argument("--output", required=True, help="output file of the pre-processed dataset")
    ap.add_argument("--type", required=True, type=str, help="amazon | movielens | amazon_ratings")
    ap.add_argument("--limit", default=None, type=int, help="read only records 0..limit of the raw dump")
    a = ap.parse_args()
    users, items = main(a.raw_dataset, a.output, a.type, a.limit)
    logging.info("%s: %d users, %d items", a.output, users, items)

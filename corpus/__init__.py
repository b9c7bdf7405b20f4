"""Seeded synthetic-corpus generator for the bftkv quorum-verification path.

Bench/test tooling: builds replica key sets, Go-shaped OpenPGP detached signatures
(SURVEY.md A.2), client certificates and ``<x,v,t,sig,ss>`` packets.  It deliberately does NOT
import ``oracle/`` (the oracle must stay an independent checker) nor the product package.
"""

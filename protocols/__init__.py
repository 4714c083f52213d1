"""Namespace package for the MI355X-native drop-in of TNO-MPC/protocols.distributed_keygen's hot path."""

"""A rendezvous port nobody else holds (fixed port numbers collide when two test processes share a box)."""
import socket


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

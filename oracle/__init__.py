"""TEST INFRASTRUCTURE: ctypes binding of the CPU restatement (oracle/libpt_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
from .binding import *  # noqa

"""Test infrastructure only: CPU restatement of the reference path (see aid_oracle.py).
Nothing under the shipped package imports this."""

"""CPU oracle for the PlaneRCNN detection hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``articulation3d_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.
"""

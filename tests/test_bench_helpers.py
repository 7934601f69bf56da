"""bench.py's host-side helpers that can be checked without a GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake_card(sysfs, card, addr, mhz, watts, vendor="0x1002", hwmon="hwmon3"):
    dev = sysfs / "devices" / "pci0000:00" / addr
    (dev / "hwmon" / hwmon).mkdir(parents=True)
    (dev / "vendor").write_text(vendor + "\n")
    (dev / "hwmon" / hwmon / "freq1_input").write_text(f"{int(mhz * 1e6)}\n")
    (dev / "hwmon" / hwmon / "power1_input").write_text(f"{int(watts * 1e6)}\n")
    (dev / "hwmon" / hwmon / "power1_cap").write_text(f"{int(1400 * 1e6)}\n")
    drm = sysfs / "class" / "drm" / card
    drm.mkdir(parents=True)
    os.symlink(dev, drm / "device")


def test_board_sample_reads_the_card_at_the_devices_pci_address(tmp_path):
    """A box shows the hwmon files of every GPU of its node; the sample must come from the one the run uses (round 5: the
    local_rank-th card was an idle neighbour: 102 MHz / 243 W beside a kernel stamping 1.86 GHz)."""
    b = _bench()
    _fake_card(tmp_path, "card0", "0000:05:00.0", 96, 240, hwmon="hwmon12")
    _fake_card(tmp_path, "card8", "0000:15:00.0", 97, 241, hwmon="hwmon13")
    _fake_card(tmp_path, "card16", "0000:65:00.0", 1900, 1050, hwmon="hwmon14")
    _fake_card(tmp_path, "card1", "0000:02:00.0", 1, 1, vendor="0x1a03")   # not an AMD GPU
    s = b.gpu_sysfs_sample(0, "0000:65:00.0", sysfs=str(tmp_path))
    assert s["matched"] and s["sclk_mhz"] == 1900 and s["power_w"] == 1050 and s["power_cap_w"] == 1400 and s["cards_visible"] == 3
    assert s["source"].endswith("hwmon14")
    # no address, or one that is not there: the local_rank-th AMD card, and the sample says it was not matched
    s = b.gpu_sysfs_sample(1, None, sysfs=str(tmp_path))
    assert not s["matched"] and s["sclk_mhz"] == 241 - 144 and s["source"].endswith("hwmon13")
    s = b.gpu_sysfs_sample(0, "0000:ff:00.0", sysfs=str(tmp_path))
    assert not s["matched"] and s["sclk_mhz"] == 96
    assert b.gpu_sysfs_sample(0, "0000:65:00.0", sysfs=str(tmp_path / "nothing")) is None


def test_energy_figures_need_a_matched_sample():
    b = _bench()
    assert b.energy_figures(None, 0.117, 64, 699050) is None
    assert b.energy_figures({"power_w": 1380.0, "matched": False}, 0.117, 64, 699050) is None
    e = b.energy_figures({"power_w": 1380.0, "matched": True, "power_of_cap": 0.986}, 0.117, 64, 699050.0)
    assert abs(e["joule_per_step"] - 0.16146) < 1e-4 and abs(e["nJ_per_channel_output"] - 3.609) < 0.01

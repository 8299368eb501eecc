! run_sph_hip_mg.f90 -- the Fortran host on several GPUs: one process per GPU, no Python, no MPI.
!
!   run_sph_hip_mg <rank> <nranks> <id_file> <ic.txt> <max_steps> [final_snapshot.txt] [sph] [saves] [tend=<end time>] [device=<d>]
!
! Every rank reads the same input file (read_data_from_file, [F]:594-716), keeps the particles of its slab along x (equal
! counts) and runs the loop body of simulate ([F]:889-920) through libsummersph_halo.so -- as run_sph_hip does on one GPU: with
! the Barnes-Hut gas self-gravity of find_forces (on the replicated tree of the all-gathered particles), sink accretion and the
! boundary cull by default, without them with "sph".  Rank 0 creates the RCCL id and writes it to <id_file> (which must not exist before);
! the other ranks wait for that file: any shared directory serves, run_mg.sh starts the ranks of one node.
! Snapshots are gathered on rank 0 in the input order, so the files are those of run_sph_hip.
! Status: never run on more than one GPU (the test pool has single-GPU boxes); with nranks = 1 it writes, byte for byte, what
! run_sph_hip writes (tests/test_halo_gpu.py).
program run_sph_hip_mg
  use, intrinsic :: iso_c_binding
  use sph_hip_binding
  use sph_hip_halo_binding
  use sph_hip_host, only: dp, particle, sink, read_data_from_file, make_save, smoothing, bounding_size
  implicit none
  character(len=512) :: id_file, filename, arg, snapshot
  type(particle), allocatable :: bodies(:), whole(:)
  type(sink), allocatable :: sinks(:)
  type(c_ptr) :: ctx, halo
  type(sph_params) :: prm
  type(sph_halo_stats) :: hs
  integer(c_int8_t) :: id(SPH_HALO_ID_BYTES)
  integer :: rank, nranks, nsteps, dev, k, n, nmine, step, save_no, io, ios, tries
  integer(c_int64_t) :: n_total
  logical :: with_saves, there, only_sph
  real(dp) :: tend, next_save
  real(c_double) :: dt, t
  real(c_double), allocatable :: xs(:), edges(:), a(:, :), s(:, :), w(:, :)
  integer(c_int64_t), allocatable :: gid(:), wgid(:)
  integer, allocatable :: owner(:)

  ctx = c_null_ptr
  halo = c_null_ptr
  if (command_argument_count() < 5) then
    write(*, *) 'usage: run_sph_hip_mg <rank> <nranks> <id_file> <ic.txt> <max_steps> [snapshot] [sph] [saves] [tend=..] [device=..]'
    error stop 2
  end if
  call get_command_argument(1, arg); read(arg, *) rank
  call get_command_argument(2, arg); read(arg, *) nranks
  call get_command_argument(3, id_file)
  call get_command_argument(4, filename)
  call get_command_argument(5, arg); read(arg, *) nsteps
  snapshot = ''
  with_saves = .false.
  only_sph = .false.
  tend = 1000.0_dp
  dev = rank
  do k = 6, command_argument_count()
    call get_command_argument(k, arg)
    if (trim(arg) == 'saves') then
      with_saves = .true.
    else if (trim(arg) == 'sph') then
      only_sph = .true.
    else if (arg(1:5) == 'tend=') then
      read(arg(6:), *) tend
    else if (arg(1:7) == 'device=') then
      read(arg(8:), *) dev
    else if (k == 6) then
      snapshot = arg
    end if
  end do

  call read_data_from_file(trim(filename), bodies, sinks)
  if (.not. allocated(bodies)) error stop 2
  n = size(bodies)

  ! slabs along x with equal counts: the same edges on every rank (same file, same sort)
  allocate(xs(n), edges(max(nranks - 1, 1)), owner(n))
  xs = bodies%position(1)
  call sort_real(xs)
  do k = 1, nranks - 1
    edges(k) = xs(min(n, int((int(k, 8) * n) / nranks) + 1))
  end do
  owner = 0
  do k = 1, nranks - 1
    where (bodies%position(1) >= edges(k)) owner = k
  end do
  nmine = count(owner == rank)

  call check(sph_params_default(prm), 'sph_params_default')
  prm%h = smoothing
  ! find_forces as it is (with the gas self-gravity term) + end-of-step accretion and cull, [F]:825,919-920
  prm%flags = ior(SPH_FLAG_SELF_GRAVITY, SPH_FLAG_ACCRETE_CULL)
  if (only_sph) prm%flags = 0
  prm%bounding_size = bounding_size
  call check(sph_ctx_create(prm, int(dev, c_int), ctx), 'sph_ctx_create')

  ! the communicator id: rank 0 makes it, the file carries it
  if (rank == 0) then
    call check(sph_halo_unique_id(id), 'sph_halo_unique_id')
    open(newunit=io, file=trim(id_file)//'.tmp', access='stream', form='unformatted', status='replace', action='write')
    write(io) id
    close(io)
    call execute_command_line('mv '//trim(id_file)//'.tmp '//trim(id_file))
  else
    tries = 0
    do
      inquire(file=trim(id_file), exist=there)
      if (there) exit
      tries = tries + 1
      if (tries > 6000) then
        write(*, *) 'rank ', rank, ': no id file after 10 minutes: ', trim(id_file)
        error stop 3
      end if
      call execute_command_line('sleep 0.1')
    end do
    open(newunit=io, file=trim(id_file), access='stream', form='unformatted', status='old', action='read')
    read(io, iostat=ios) id
    close(io)
    if (ios /= 0) error stop 3
  end if
  call check(sph_halo_create(ctx, id, int(rank, c_int32_t), int(nranks, c_int32_t), halo), 'sph_halo_create')
  call check(sph_halo_set_slabs(halo, edges, 32_c_int32_t), 'sph_halo_set_slabs')

  allocate(s(size(sinks), 7))
  s(:, 1) = sinks%position(1); s(:, 2) = sinks%position(2); s(:, 3) = sinks%position(3)
  s(:, 4) = sinks%velocity(1); s(:, 5) = sinks%velocity(2); s(:, 6) = sinks%velocity(3)
  s(:, 7) = sinks%mass
  call check(sph_set_sinks(ctx, int(size(sinks), c_int32_t), s(:, 1), s(:, 2), s(:, 3), s(:, 4), s(:, 5), s(:, 6), s(:, 7)), &
             'sph_set_sinks')
  call check(sph_set_sink_radii(ctx, int(size(sinks), c_int32_t), sinks%radius), 'sph_set_sink_radii')

  allocate(a(max(nmine, 1), 9), gid(max(nmine, 1)))
  nmine = 0
  do k = 1, n
    if (owner(k) /= rank) cycle
    nmine = nmine + 1
    a(nmine, 1:3) = bodies(k)%position
    a(nmine, 4:6) = bodies(k)%velocity
    a(nmine, 7) = bodies(k)%internal_energy
    a(nmine, 8) = bodies(k)%mass
    a(nmine, 9) = bodies(k)%alpha
    gid(nmine) = k - 1
  end do
  call check(sph_halo_upload(halo, int(nmine, c_int64_t), a(:, 1), a(:, 2), a(:, 3), a(:, 4), a(:, 5), a(:, 6), a(:, 7), &
                             a(:, 8), a(:, 9), gid), 'sph_halo_upload')

  allocate(w(n, 9), wgid(n), whole(n))
  t = 0.0_c_double
  dt = 1.0e-2_c_double
  next_save = 0.0_dp
  save_no = 0
  step = 0
  if (rank == 0) write(*, '(A,I0,1X,ES25.17E3)') 'dt ', 0, dt
  do while (t < tend .and. step < nsteps)
    if (with_saves) then
      if (save_no == 0 .or. t > next_save) then       ! the reference's cadence, as in sph_hip_host
        call collect()
        if (rank == 0) call make_save(whole(1:int(n_total)), sinks, save_no)
        save_no = save_no + 1
        next_save = (save_no * tend) / 1000
      end if
    end if
    call check(sph_halo_run(halo, 1_c_int32_t, dt, t), 'sph_halo_run')
    step = step + 1
    if (rank == 0) write(*, '(A,I0,1X,ES25.17E3)') 'dt ', step, dt
  end do
  if (len_trim(snapshot) > 0) then
    call collect()
    if (rank == 0) call make_save(whole(1:int(n_total)), sinks, 0, trim(snapshot))
  end if
  call check(sph_halo_get_stats(halo, hs), 'sph_halo_get_stats')
  write(*, '(A,I0,A,I0,A,I0,A,I0,A,I0,A,I0)') 'rank ', rank, ': owned ', sph_halo_count(halo), ' ghosts ', hs%ghosts, &
    ' migrated(all ranks) ', hs%migrated, ' exchanges ', hs%exchanges, ' accreted+culled ', hs%removed
  call check(sph_halo_destroy(halo), 'sph_halo_destroy')
  call check(sph_ctx_destroy(ctx), 'sph_ctx_destroy')

contains

  subroutine check(status, what)
    integer(c_int), intent(in) :: status
    character(len=*), intent(in) :: what
    if (status /= SPH_OK) then
      write(*, *) 'rank ', rank, ': ', what, ' failed: ', c_message(sph_strerror(status)), ' -- ', &
        c_message(sph_halo_last_error(halo)), ' -- ', c_message(sph_last_error(ctx))
      error stop 1
    end if
  end subroutine check

  ! every rank's particles on rank 0, in input order (+ the sinks, which every rank holds identically)
  subroutine collect()
    real(c_double), allocatable :: sg(:, :)
    integer :: i, ns
    call check(sph_halo_gather_root(halo, 0_c_int32_t, int(n, c_int64_t), n_total, w(:, 1), w(:, 2), w(:, 3), w(:, 4), w(:, 5), &
                                    w(:, 6), w(:, 7), w(:, 8), w(:, 9), wgid), 'sph_halo_gather_root')
    if (rank /= 0) return
    do i = 1, int(n_total)
      whole(i)%position = w(i, 1:3)
      whole(i)%velocity = w(i, 4:6)
      whole(i)%internal_energy = w(i, 7)
      whole(i)%mass = w(i, 8)
      whole(i)%alpha = w(i, 9)
      whole(i)%number = int(wgid(i)) + 1
    end do
    ns = size(sinks)
    allocate(sg(ns, 10))
    call check(sph_get_sinks(ctx, int(ns, c_int32_t), sg(:, 1), sg(:, 2), sg(:, 3), sg(:, 4), sg(:, 5), sg(:, 6), sg(:, 7), &
                             sg(:, 8), sg(:, 9), sg(:, 10)), 'sph_get_sinks')
    do i = 1, ns
      sinks(i)%position = sg(i, 1:3)
      sinks(i)%velocity = sg(i, 4:6)
      sinks(i)%mass = sg(i, 7)
    end do
  end subroutine collect

  ! in-place heap sort
  subroutine sort_real(v)
    real(c_double), intent(inout) :: v(:)
    integer :: m, i, parent, child
    real(c_double) :: tmp
    m = size(v)
    do i = m / 2, 1, -1
      call sift(i, m)
    end do
    do i = m, 2, -1
      tmp = v(1); v(1) = v(i); v(i) = tmp
      call sift(1, i - 1)
    end do
  contains
    subroutine sift(start, last)
      integer, intent(in) :: start, last
      parent = start
      do
        child = 2 * parent
        if (child > last) exit
        if (child < last) then
          if (v(child + 1) > v(child)) child = child + 1
        end if
        if (v(parent) >= v(child)) exit
        tmp = v(parent); v(parent) = v(child); v(child) = tmp
        parent = child
      end do
    end subroutine sift
  end subroutine sort_real
end program run_sph_hip_mg
